"""The N>1 exchange step on CPU: world_size-2 gloo processes running the same reduce code bench.py uses."""
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

from sparksmithwaterman_amd import distributed as swd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    totals_all = [10, 694, 237, 694, 0, 5, 694, 300]      # 8 references, three tied winners
    lo, hi = swd.shard_bounds(len(totals_all), rank, world)
    best, winners = swd.global_max_with_ties(totals_all[lo:hi], range(lo, hi))
    topk = swd.global_top_k(totals_all[lo:hi], range(lo, hi), 4)
    neg = swd.global_max_with_ties([-3, -1], [lo, lo + 1])     # `int max = 0`: negative totals never win
    red = swd.MaxReducer("cpu")                                 # the preallocated-buffer form bench.py uses over RCCL
    assert red(totals_all[lo:hi], list(range(lo, hi))) == (best, winners)
    assert red([-3, -1], [lo, lo + 1]) == (0, [])
    # pipelined form: two exchanges in flight, collected in order (what bench.py's step loop does)
    t1 = red.submit(totals_all[lo:hi], list(range(lo, hi)))
    t2 = red.submit([7, 7] if rank == 0 else [7, 3], [lo, lo + 1])
    assert red.collect(t1) == (best, winners)
    assert red.collect(t2) == (7, [0, 1, 4])
    try:
        red.collect(t1 - 5)
        assert False, "a ticket that left the ring must be refused"
    except ValueError:
        pass
    # more tied winners than the fixed payload carries: duplicated references ...
    many = list(range(1000 * rank, 1000 * rank + 150))
    full = swd.global_max_with_ties([42] * 150, many)
    assert full == (42, sorted(list(range(0, 150)) + list(range(1000, 1150))))
    assert red([42] * 150, many) == full
    # ... only one rank over the cap, the other one below the maximum ...
    if rank == 0:
        assert red([9] * 100, many[:100]) == (9, many[:100])
    else:
        assert red([3, 8], [1000, 1001]) == (9, list(range(0, 100)))
    # ... and the all-zero case: `int max = 0` makes EVERY reference of a shard where nothing scores a winner
    zero = swd.global_max_with_ties([0] * 70, list(range(100 * rank, 100 * rank + 70)))
    assert zero == (0, list(range(0, 70)) + list(range(100, 170)))
    q.put((rank, best, winners, topk, neg))
    dist.destroy_process_group()


def test_gloo_world2_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, best, winners, topk, neg in res:
        assert best == 694 and winners == [1, 3, 6]
        assert topk == [(694, 1), (694, 3), (694, 6), (300, 7)]
        assert neg == (0, [])


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1000, 1001):
        for w in (1, 2, 3, 8):
            spans = [swd.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_single_process_reduce_without_init():
    assert swd.global_max_with_ties([1, 9, 9], [10, 11, 12]) == (9, [11, 12])
    assert swd.global_top_k([1, 9, 9], [10, 11, 12], 2) == [(9, 11), (9, 12)]


def test_shard_by_length_is_a_balanced_partition():
    import numpy as np
    rng = np.random.default_rng(7)
    lens = np.clip(np.rint(np.exp(rng.normal(7.38, 0.77, 5003))), 50, 100000).astype(np.int64)
    for w in (1, 2, 3, 8):
        parts = [swd.shard_by_length(lens, r, w) for r in range(w)]
        assert sorted(np.concatenate(parts).tolist()) == list(range(lens.size))
        assert all((np.diff(p) > 0).all() for p in parts)                       # ascending global ids
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        cells = [int(lens[p].sum()) for p in parts]
        assert max(cells) - min(cells) <= int(lens.max())                        # within one (longest) reference of each other
    assert swd.shard_by_length([], 0, 2).size == 0


def test_top_k_orders_by_total_then_id_and_keeps_negative_totals():
    assert swd.global_top_k([5, -7, 5, 0, -1], [40, 41, 12, 3, 9], 4) == [(5, 12), (5, 40), (0, 3), (-1, 9)]
    assert swd.global_top_k([], [], 3) == []
    assert swd.global_top_k([2**31 - 1, -2**31], [0, 2**32 - 2], 5) == [(2**31 - 1, 0), (-2**31, 2**32 - 2)]
