"""Reference-sharded multi-GPU layer: one process per GPU, torch.distributed for the only exchange step.

The reference data-parallelises over references (`sc.parallelize(list).mapToPair(new MapRef())`,
src/sw/Distribution.java:337-338) and then reduces to the best total(s) on the driver (:341-353; control-path
semantics :600-613).  Here every rank aligns its own shard of references against the full read set with no
data-path collective; the reduce is an all-reduce(max) of one int32 plus an all-gather of the ranks' few
winners -- bytes over xGMI, latency-bound by design.  Backend "nccl" is RCCL on ROCm; "gloo" in CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_references(refs, rank, world):
    lo, hi = shard_bounds(len(refs), rank, world)
    return refs[lo:hi], lo


def global_max_with_ties(local_totals, global_ids, device=None, group=None, cap=64):
    """Control-path reduce across ranks (Distribution.java:600-613): returns (max_total, sorted ids of every
    reference whose total equals it).  local_totals/global_ids: equal-length int sequences of this rank's shard.
    `max` starts at 0 like the reference's (`int max = 0`, :573), so totals below 0 never win."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    t = torch.as_tensor(list(local_totals), dtype=torch.int64)
    ids = torch.as_tensor(list(global_ids), dtype=torch.int64)
    local_best = int(t.max()) if t.numel() else 0
    best = torch.tensor([max(local_best, 0)], dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(best, op=dist.ReduceOp.MAX, group=group)
    gbest = int(best.item())
    mine = ids[t == gbest][:cap] if t.numel() else ids[:0]
    payload = torch.full((cap + 1,), -1, dtype=torch.int64, device=device)
    payload[0] = mine.numel()
    if mine.numel():
        payload[1:1 + mine.numel()] = mine.to(payload.device)
    if world > 1:
        gathered = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(gathered, payload, group=group)
    else:
        gathered = [payload]
    winners = []
    for g in gathered:
        g = g.cpu()
        winners.extend(int(x) for x in g[1:1 + int(g[0])])
    return gbest, sorted(winners)


def global_top_k(local_totals, global_ids, k, device=None, group=None):
    """Top-k (total, id) over all ranks: each rank contributes its local top-k (k x 16 B), merged everywhere.
    Ties are broken by ascending reference id so every rank returns the same list."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    t = torch.as_tensor(list(local_totals), dtype=torch.int64)
    ids = torch.as_tensor(list(global_ids), dtype=torch.int64)
    order = sorted(range(t.numel()), key=lambda i: (-int(t[i]), int(ids[i])))[:k]
    payload = torch.full((k, 2), -(1 << 62), dtype=torch.int64, device=device)
    for r, i in enumerate(order):
        payload[r, 0] = t[i]
        payload[r, 1] = ids[i]
    if world > 1:
        gathered = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(gathered, payload, group=group)
    else:
        gathered = [payload]
    rows = []
    for g in gathered:
        for tot, i in g.cpu().tolist():
            if tot > -(1 << 62):
                rows.append((int(tot), int(i)))
    rows.sort(key=lambda x: (-x[0], x[1]))
    return rows[:k]
