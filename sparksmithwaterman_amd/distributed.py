"""Reference-sharded multi-GPU layer: one process per GPU, torch.distributed for the only exchange step.

The reference data-parallelises over references (`sc.parallelize(list).mapToPair(new MapRef())`,
src/sw/Distribution.java:337-338) and then reduces to the best total(s) on the driver (:341-353; control-path
semantics :600-613).  Here every rank aligns its own shard of references against the full read set with no
data-path collective; the reduce is one all-gather of every rank's {local max, its few winners} -- bytes over
xGMI, latency-bound by design.  Backend "nccl" is RCCL on ROCm; "gloo" in CPU tests.
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


def shard_by_length(lengths, rank, world):
    """Length-balanced shard of ONE reference set: the references sorted by length, longest first, are
    dealt to the ranks in a snake (0..w-1, w-1..0, ...), so every rank gets the same number of references (+-1) and
    nearly the same number of DP cells whatever the length distribution (NCBI-shaped sets are log-normal: a contiguous
    split by count can leave one rank with the few 100 kbp records).  Returns this rank's GLOBAL reference ids, ascending
    (the order the references had in the set, which is what the result file keeps, Distribution.java:359)."""
    import numpy as np
    n = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-n, kind="stable")              # longest first, ties in set order
    pos = np.arange(order.size)
    lap, col = pos // world, pos % world
    owner = np.where(lap % 2 == 0, col, world - 1 - col)
    return np.sort(order[owner == rank])


def _decode(g, cap):
    """rows {local max, uncapped count, ids...} of every rank -> (global max, winners or None when a winning rank has
    more ids than fit the fixed payload, the largest such count)"""
    gbest = int(g[:, 0].max())
    winners, need = [], 0
    for row in g:
        if int(row[0]) == gbest:
            cnt = int(row[1])
            need = max(need, cnt)
            winners.extend(int(x) for x in row[2:2 + min(cnt, cap)])
    return gbest, (sorted(winners) if need <= cap else None), need


def _gather_all_winners(mine, is_winner, need, device, group):
    """second exchange, only when some rank holds more tied winners than the fixed payload carries (duplicated
    references, or a shard where nothing scores: `int max = 0` makes every total-0 reference a winner,
    Distribution.java:573,600-613): every rank contributes `need` slots."""
    import numpy as np
    world = dist.get_world_size(group)
    pay = np.full(need + 1, -1, dtype=np.int64)
    if is_winner:
        pay[0] = mine.size
        pay[1:1 + mine.size] = mine
    else:
        pay[0] = 0
    payload = torch.from_numpy(pay)
    if device is not None:
        payload = payload.to(device)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    out = []
    for row in torch.stack(gathered).cpu().numpy():
        out.extend(int(x) for x in row[1:1 + int(row[0])])
    return sorted(out)


def global_max_with_ties(local_totals, global_ids, device=None, group=None, cap=64):
    """Control-path reduce across ranks (Distribution.java:600-613): returns (max_total, sorted ids of EVERY
    reference whose total equals it).  local_totals/global_ids: equal-length int sequences (or numpy arrays) of
    this rank's shard.  `max` starts at 0 like the reference's (`int max = 0`, :573), so totals below 0 never win.

    ONE collective in the usual case: every rank contributes {its local max, how many ids reach it, the first `cap` of
    them} (cap+2 int64 = 528 B) to an all-gather; the global max and its references follow locally and identically on
    every rank.  Only when a winning rank holds more than `cap` ids does a second all-gather, sized by the largest
    count, fetch the complete lists -- nothing is ever cut."""
    import numpy as np
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    t = np.asarray(local_totals, dtype=np.int64)
    ids = np.asarray(global_ids, dtype=np.int64)
    local_best = max(int(t.max()), 0) if t.size else 0
    mine = ids[t == local_best] if t.size else ids[:0]
    if world == 1:
        return local_best, sorted(int(x) for x in mine)
    pay = np.full(cap + 2, -1, dtype=np.int64)
    pay[0] = local_best
    pay[1] = mine.size
    pay[2:2 + min(mine.size, cap)] = mine[:cap]
    payload = torch.from_numpy(pay)
    if device is not None:
        payload = payload.to(device)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    g = torch.stack(gathered).cpu().numpy()
    gbest, winners, need = _decode(g, cap)
    if winners is None:
        winners = _gather_all_winners(mine, local_best == gbest, need, device, group)
    return gbest, winners


class MaxReducer:
    """The same reduce as global_max_with_ties with every buffer allocated once: per call one pinned->device
    copy, one all_gather_into_tensor (RCCL), one device->pinned copy, one event wait.  For the per-step reduce of
    bench.py, where the collective's latency is all there is to pay.

    submit() only enqueues the exchange and returns a ticket; collect(ticket) waits for it and decodes.  A driver
    that streams shards (bench.py) submits step k and collects step k-1, so the collective's latency hides behind
    the next shard's kernels (the library runs on its own HIP stream).  `depth` exchanges may be in flight.
    More than `cap` tied winners on a winning rank: collect() runs the second, exactly-sized exchange (every rank sees
    the same counts, so every rank enters it)."""

    def __init__(self, device, cap=64, group=None, depth=2, always_exchange=False):
        self.cap, self.group, self.device, self.depth = cap, group, device, depth
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchange = self.world > 1 or (always_exchange and dist.is_initialized())   # (1-rank groups: rehearsal only)
        on_gpu = torch.device(device).type == "cuda"
        self.on_gpu = on_gpu
        self.flat_gather = dist.is_initialized() and dist.get_backend(group) != "gloo"
        self.slots = []
        for _ in range(depth):
            h_pay = torch.empty(cap + 2, dtype=torch.int64, pin_memory=on_gpu)
            h_all = torch.empty(self.world * (cap + 2), dtype=torch.int64, pin_memory=on_gpu)
            self.slots.append({
                "h_pay": h_pay, "pay_np": h_pay.numpy(),
                "d_pay": torch.empty(cap + 2, dtype=torch.int64, device=device),
                "d_all": torch.empty(self.world * (cap + 2), dtype=torch.int64, device=device),
                "h_all": h_all, "all_np": h_all.numpy().reshape(self.world, cap + 2),
                "event": torch.cuda.Event() if on_gpu else None, "local": None, "mine": None,
            })
        self.n_submitted = 0

    def submit(self, local_totals, global_ids):
        import numpy as np
        t = np.asarray(local_totals)
        local_best = max(int(t.max()), 0) if t.size else 0
        mine = np.asarray(global_ids, dtype=np.int64)[t == local_best] if t.size else np.empty(0, dtype=np.int64)
        ticket = self.n_submitted
        self.n_submitted += 1
        sl = self.slots[ticket % self.depth]
        sl["local"] = (local_best, sorted(int(x) for x in mine))
        sl["mine"] = mine
        if not self.exchange:
            return ticket
        sl["pay_np"][0] = local_best
        sl["pay_np"][1] = mine.size
        k = min(mine.size, self.cap)
        sl["pay_np"][2:2 + k] = mine[:k]
        sl["d_pay"].copy_(sl["h_pay"], non_blocking=True)
        if self.flat_gather:
            dist.all_gather_into_tensor(sl["d_all"], sl["d_pay"], group=self.group)
        else:       # gloo (CPU tests) has no flat-tensor all-gather
            dist.all_gather(list(sl["d_all"].view(self.world, self.cap + 2).unbind(0)), sl["d_pay"], group=self.group)
        sl["h_all"].copy_(sl["d_all"], non_blocking=True)
        if sl["event"] is not None:
            sl["event"].record()
        return ticket

    def collect(self, ticket):
        if ticket < self.n_submitted - self.depth or ticket >= self.n_submitted:
            raise ValueError("ticket %d is not in flight" % ticket)
        sl = self.slots[ticket % self.depth]
        if not self.exchange:
            return sl["local"]
        if sl["event"] is not None:
            sl["event"].synchronize()
        gbest, winners, need = _decode(sl["all_np"], self.cap)
        if winners is None:
            winners = _gather_all_winners(sl["mine"], sl["local"][0] == gbest, need,
                                          self.device if self.on_gpu else None, self.group)
        return gbest, winners

    def __call__(self, local_totals, global_ids):
        return self.collect(self.submit(local_totals, global_ids))


def global_top_k(local_totals, global_ids, k, device=None, group=None):
    """Top-k (total, id) over all ranks: each rank contributes its local top-k (k x 16 B), merged everywhere.
    Ties are broken by ascending reference id so every rank returns the same list.

    Both selections are ONE torch.topk over a composite int64 key, total * 2^32 + (2^32 - 1 - id): totals are Java
    ints (Distribution.java:424) and ids fit 32 bits, so the key orders by total descending, then id ascending."""
    import numpy as np
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(local_totals, dtype=np.int64)))
    ids = torch.from_numpy(np.ascontiguousarray(np.asarray(global_ids, dtype=np.int64)))
    if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) > 0xFFFFFFFF):
        raise ValueError("reference ids must fit 32 bits")
    key = (t << 32) + (0xFFFFFFFF - ids)
    sentinel = -(1 << 63)                              # (below every real key but {total -2^31, id 2^32 - 1})
    payload = torch.full((k,), sentinel, dtype=torch.int64)
    kk = min(k, key.numel())
    if kk:
        payload[:kk] = torch.topk(key, kk).values
    if device is not None:
        payload = payload.to(device)
    if world > 1:
        gathered = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(gathered, payload, group=group)
        allk = torch.cat(gathered).cpu()
    else:
        allk = payload.cpu()
    allk = allk[allk > sentinel]
    best = torch.topk(allk, min(k, allk.numel())).values if allk.numel() else allk
    tot = best >> 32                                   # (arithmetic shift: negative totals survive)
    rid = 0xFFFFFFFF - (best & 0xFFFFFFFF)
    return [(int(a), int(b)) for a, b in zip(tot.tolist(), rid.tolist())]
