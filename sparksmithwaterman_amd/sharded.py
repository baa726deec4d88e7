"""ONE reference set sharded over the ranks of a node (BASELINE.json configs[3]): align, then the driver's reduce.

The reference repartitions one reference list over its executors (`sc.parallelize(list).mapToPair(new MapRef())`,
src/sw/Distribution.java:337-338) and reduces to the best total(s) on the driver (:341-353; control semantics :600-613).
Here rank r of `world` takes its length-balanced shard of the SAME set (distributed.shard_by_length),
aligns it against every read on its own GPU with no data-path collective, and the only exchange is the reduce:
max-with-ties (one all-gather of {local max, winners}) and top-K (one all-gather of K composite keys).  Backend "nccl"
is RCCL over xGMI; "gloo" when several ranks rehearse on one GPU and in the CPU tests.

    python -m sparksmithwaterman_amd.sharded --n-refs 2000 --n-reads 32 --world 2 --out /tmp/x   (starts its own ranks)
"""
import json
import os
import time

import numpy as np

from . import distributed as swd


def align_shard(ctx, refs, reads, params, local_ids, stream_chunk_bytes=0, slots=0):
    """Per-reference totals (Distribution.java:424) of this rank's references `local_ids` of the set `refs`.
    Small shards go through one batch; `stream_chunk_bytes` > 0 streams the shard in chunks of that many sequence bytes
    (swmi_stream_*: parse/upload of chunk k+1 overlaps the kernels of chunk k) -- what a shard of 12,500 references x
    10,000 reads needs, whose pairs do not fit one batch's bookkeeping.  Returns (totals int32[len(local_ids)], cells)."""
    mine = [refs[int(i)] for i in local_ids]
    cells = sum(len(r) for r in mine) * sum(len(q) for q in reads)
    if not mine:
        return np.zeros(0, dtype=np.int32), 0
    if stream_chunk_bytes:
        st = ctx.stream(reads, params, slots=slots, chunk_bytes=stream_chunk_bytes)
        try:
            st.push(mine).finish()
            totals = st.totals().copy()
        finally:
            st.close()
        return totals, cells
    b = ctx.upload(mine, reads)
    try:
        b.run(params)
        totals = b.ref_totals().copy()
    finally:
        b.free()
    return totals, cells


def run_sharded(ctx, refs, reads, params=None, rank=0, world=1, top_k=8, reduce_device=None, group=None,
                stream_chunk_bytes=0, slots=0):
    """Shard `refs` by length, align this rank's shard against all `reads`, reduce over the ranks.
    Returns a dict: best / winners (max-with-ties, global reference ids), top_k [(total, id)], this rank's ids and totals,
    and the wall times of the two phases (align_s, reduce_s)."""
    from . import aligner
    params = params if params is not None else aligner.make_params()
    lengths = np.fromiter((len(r) for r in refs), dtype=np.int64, count=len(refs))
    local_ids = swd.shard_by_length(lengths, rank, world)
    t0 = time.perf_counter()
    totals, cells = align_shard(ctx, refs, reads, params, local_ids, stream_chunk_bytes, slots)
    t1 = time.perf_counter()
    best, winners = swd.global_max_with_ties(totals, local_ids, device=reduce_device, group=group)
    top = swd.global_top_k(totals, local_ids, top_k, device=reduce_device, group=group)
    t2 = time.perf_counter()
    return {"best": int(best), "winners": [int(x) for x in winners], "top_k": [(int(a), int(b)) for a, b in top],
            "local_ids": local_ids, "local_totals": totals, "cells": int(cells), "align_s": t1 - t0, "reduce_s": t2 - t1}


def _rank_main(args):
    """one rank of the CLI: every rank generates the same synthetic set (deterministic), takes its shard, runs, and rank 0
    writes the reduced result; every rank writes its own totals (the test checks them against the oracle)."""
    import torch
    import torch.distributed as dist
    import sparksmithwaterman_amd as sw
    from . import synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n_dev = torch.cuda.device_count()                     # (does not initialise the GPU)
    one_gpu = n_dev < world or os.environ.get("SWMI_ONE_GPU") == "1"
    dev_id = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", rank))
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(dev_id)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_id))
    refs, reads = synth.config_multi_read(args.n_refs, args.n_reads, seed=args.seed)
    ctx = sw.Context(dev_id)
    res = run_sharded(ctx, refs, reads, None, rank, world, args.top_k,
                      reduce_device=None if (one_gpu or world == 1) else torch.device("cuda", dev_id),
                      stream_chunk_bytes=args.stream_chunk_bytes)
    ctx.close()
    out = {"rank": rank, "world": world, "best": res["best"], "winners": res["winners"], "top_k": res["top_k"],
           "local_ids": [int(x) for x in res["local_ids"]], "local_totals": [int(x) for x in res["local_totals"]],
           "cells": res["cells"], "align_s": res["align_s"], "reduce_s": res["reduce_s"],
           "backend": "none" if world == 1 else ("gloo" if one_gpu else "nccl")}
    with open(os.path.join(args.out, "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    import argparse
    import socket
    import subprocess
    import sys
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-refs", type=int, default=2000)
    ap.add_argument("--n-reads", type=int, default=32)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--top-k", type=int, default=8)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--stream-chunk-bytes", type=int, default=0)
    ap.add_argument("--out", required=True, help="directory for rank<r>.json")
    args = ap.parse_args(argv)
    if "RANK" in os.environ:
        _rank_main(args)
        return 0
    # launcher: this process makes no GPU call; the ranks are fresh children (never a re-exec of a GPU process)
    os.makedirs(args.out, exist_ok=True)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.world))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-m", "sparksmithwaterman_amd.sharded"] + (argv if argv is not None else sys.argv[1:]),
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(args.world)]
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
