#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Smith-Waterman hot path on N MI355X (contract in the task brief).

A "step" is one pass of the hot path over one batch: score sweep + tied-maximum lists + device traceback for every
(reference, read) pair, compact result records written to host memory, and (N > 1) the max/top-K reduce over RCCL.
Inputs are resident in HBM before the timed region starts.  Steps are independent of each other, so two are kept in flight per
GPU (--in-flight, default 2), each a full pass over its own resident copy of the batch with its own context, stream, workspace
and pinned result block: the sweep of one runs beside the tail of the other's traceback launch.  `value` / `ms_per_step` are that
loop's; `gcups_one_in_flight` / `ms_per_step_one_in_flight` the same steps strictly one after the other.

N = 1 workload = BASELINE.json configs[1]: one 150 bp read x 1,000 synthetic 2 kbp references (3.0e8 cells,
SplitMix64 seed 1).  N > 1: every rank holds its own 1,000-reference shard (weak scaling, references sharded
as the reference's `parallelize(refs)` does); value = cells of ALL ranks / max-over-ranks time.

`python bench.py --gpus N` (N > 1) without WORLD_SIZE in the environment starts the N rank processes itself (fresh
children of a parent that never touches the GPU); under `torch.distributed.run` it is one of the ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
EVENT_EVERY = 4         # the timed steps whose sweep kernel is bracketed by HIP events: every 4th (bench.py: timed loop)

WORKLOAD_OF_MODE = {
    1: "score-only sweep (lane-state checkpoints + one maximum per 32-step window) + traceback by window re-sweep "
       "(tied maxima listed, alignments walked) + result records written to host memory",
    2: "score-only sweep (checkpoints, tied maxima tracked by events) + traceback by window re-sweep + result records",
    0: "sweep writing the 2-bit direction field to HBM + tied maxima + traceback over the field + result records",
}


def alg_bytes(m, n):
    """packed ref + packed read + 2-bit direction field + traceback re-read + record."""
    return -(-n // 4) + -(-m // 4) + -(-(m * n) // 4) + -(-(m + n) // 4) + 16


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50; 2 with --scaling strong)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 5; 1 with --scaling strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): configs[1], every rank its own 1,000-reference shard.  strong: configs[3], ONE set of --n-refs "
                         "NCBI-shaped references x --n-reads reads sharded by length over the ranks, then the max/top-K reduce")
    ap.add_argument("--n-refs", type=int, default=None, help="references (default 1000; 100000 with --scaling strong)")
    ap.add_argument("--n-reads", type=int, default=1000, help="--scaling strong: reads (BASELINE.json configs[3] names 10000)")
    ap.add_argument("--chunk-kb", type=int, default=512, help="--scaling strong: KiB of reference bases per streamed chunk")
    ap.add_argument("--top-k", type=int, default=8)
    ap.add_argument("--ref-len", type=int, default=2000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight per GPU (one context, stream and host thread each): 2 lets step k+1's sweep fill the tail "
                         "of step k's traceback launch; 1 = strictly one step after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--d2h-copy", action="store_true", help="fetch results with a D2H copy instead of zero-copy writes to pinned memory")
    ap.add_argument("--col-chunks", type=int, default=None, help="column chunks per pair (include/swmi.h): 0 automatic, 1 never, N force")
    ap.add_argument("--tfused", type=int, default=None, help="transposed fused kernel (include/swmi.h): -1 automatic, 0 never, 1 every pair that qualifies")
    ap.add_argument("--mode", type=int, default=None, help="kernel pipeline (include/swmi.h): 1 default, 2 event-tracked maxima, 0 HBM direction field")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="any swmi_set_option knob, e.g. device_strings=0 (A/B runs)")
    args = ap.parse_args()
    strong = args.scaling == "strong"
    if args.steps is None:
        args.steps = 2 if strong else 50
    if args.warmup is None:
        args.warmup = 1 if strong else 5
    if args.n_refs is None:
        args.n_refs = 100000 if strong else 1000
    return args


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """--gpus N with no rendezvous in the environment: this process becomes a launcher.  It makes NO GPU call (it does
    not even import torch): the ranks are fresh children, never a re-exec of a process that touched the GPU.
    Rank 0 prints the JSON line on the inherited stdout; a failed rank ends the others and the exit code is non-zero."""
    n = args.gpus
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(free_port()))
    env["WORLD_SIZE"] = str(n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env)
        e["RANK"] = str(r)
        e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:          # the exact processes this launcher started, nothing else
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc)


def effective_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a 16-core share of a 256-thread host)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except Exception:
        pass
    return cores


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)          # never returns

    import numpy as np
    import torch
    import torch.distributed as dist
    import sparksmithwaterman_amd as sw
    from sparksmithwaterman_amd import synth
    from sparksmithwaterman_amd import distributed as swd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a box with fewer GPUs than ranks (torch.cuda.device_count() does not initialise the GPU): every rank
    # on GPU 0 and gloo for the reduce.  SWMI_BENCH_ONE_GPU=1 forces it.
    n_dev = torch.cuda.device_count()
    one_gpu = world > 1 and (os.environ.get("SWMI_BENCH_ONE_GPU") == "1" or n_dev < world)
    if one_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    # rehearsal of the multi-GPU step loop on ONE GPU: a one-rank nccl (RCCL) group with the exchange forced on, so that
    # the helper thread issues real RCCL collectives next to the library's kernels (SWMI_BENCH_REHEARSE_RCCL=1)
    rehearse_rccl = world == 1 and os.environ.get("SWMI_BENCH_REHEARSE_RCCL") == "1"
    if rehearse_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cpu") if (one_gpu and world > 1) else torch.device("cuda", local_rank)
    if args.scaling == "strong":
        strong_scaling(args, world, rank, local_rank, one_gpu, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- synthetic inputs: shard `rank` of a world*n_refs reference set, the read of shard 0 --------------
    refs0, reads = synth.config_1k(args.n_refs, args.ref_len, args.read_len, seed=1)
    refs = refs0 if rank == 0 else synth.config_1k(args.n_refs, args.ref_len, args.read_len, seed=1 + 7919 * rank)[0]
    id0 = rank * args.n_refs
    m = len(reads[0])
    cells_rank = sum(len(r) for r in refs) * m
    bytes_rank = sum(alg_bytes(m, len(r)) for r in refs)

    # Steps are passes over a batch whose inputs are resident in HBM.  --in-flight 2 (default) keeps TWO of them going, each with
    # its own context (stream, host thread, workspace, pinned result block), the way a driver feeding partition after partition
    # would: the traceback launch of step k ends with a few slow pairs on an almost empty chip (mean workgroup 54 k ticks,
    # slowest 100 k), and the sweep of step k+1 -- one wavefront per SIMD -- runs in that tail.  Every step still does all of its
    # work and delivers its results to host memory inside the timed region.
    depth = max(1, args.in_flight)

    def make_ctx():
        c = sw.Context(local_rank)
        # HIP events around the SWEEP of every EVENT_EVERY-th timed step (roofline.achieved needs that kernel's duration live,
        # inside the timed region); the traceback's duration is measured with the full set of events in a separate, labelled
        # loop outside it -- every marker packet costs the step ~3.5 us (profiles/r03/host_breakdown_headline.txt)
        c.set_option("profiling", 2)
        if args.mode is not None:
            c.set_option("mode", args.mode)
        if args.col_chunks is not None:
            c.set_option("col_chunks", args.col_chunks)
        if args.tfused is not None:
            c.set_option("tfused", args.tfused)
        if args.d2h_copy:
            c.set_option("zero_copy", 0)
        for kv in args.opt:
            name, _, value = kv.partition("=")
            c.set_option(name, int(value))
        return c

    ctxs = [make_ctx() for _ in range(depth)]
    batches = [c.upload(refs, reads) for c in ctxs]          # H2D happens here, outside the timed region
    ctx, batch = ctxs[0], batches[0]
    params = sw.make_params()
    for b in batches:
        b.run(params)
    mode = batch.pipeline_mode()             # the pipeline the library chose for this batch (or the one forced above)
    kernel_name = {0: "sw_fill_kernel", 1: "sw_sweep_winmax_kernel", 2: "sw_fill_score_kernel"}[mode]
    tfused = batch.timing().tfused_pairs == len(refs) * len(reads)      # option --tfused 1: sweep AND traceback in one launch
    if tfused:
        kernel_name = "sw_tfused_kernel (sweep + traceback of every pair in one launch)"

    gids = np.arange(id0, id0 + len(refs), dtype=np.int64)

    reducer = None
    if world > 1 or rehearse_rccl:
        # buffers allocated once; one RCCL all-gather per step, collected three steps later: a rank that is briefly late does
        # not stall the others' step loops (every exchange of the timed steps still completes inside the timed region)
        reducer = swd.MaxReducer("cpu" if one_gpu else dev, always_exchange=rehearse_rccl, depth=4)

    # The path's one exchange step (max total + its references) is done for shard k-1 while the GPU works on shard k,
    # the way a driver streaming shards would: the run is started on the library's own host thread
    # (swmi_batch_run_async), this thread submits the exchange of the previous shard's totals and collects the one
    # before, then waits for the run.  Every exchange of the timed steps has completed before the timed region ends.
    pending = []               # tickets of submitted, not yet collected exchanges
    last_result = [None]
    order = []                 # slots in flight, oldest first
    sampled_slot = [False] * depth
    acc = {"fill_ms": 0.0, "launches": 0, "timed_steps": 0}

    def retire(i):
        """completes the run of slot i: its results are in host memory; returns its totals for the exchange"""
        if depth > 1:
            batches[i].wait()
        if sampled_slot[i]:
            t = batches[i].timing()
            acc["fill_ms"] += t.fill_ms; acc["launches"] += t.fill_launches; acc["timed_steps"] += 1
            sampled_slot[i] = False
        return batches[i].ref_totals() if reducer is not None else None

    def exchange(totals):
        """the path's one exchange step for a retired shard (submitted now, collected one step later)"""
        if totals is not None:
            pending.append(reducer.submit(totals, gids))
            if len(pending) > 3:
                last_result[0] = reducer.collect(pending.pop(0))

    def step(k, timed=False):
        i = k % depth
        totals = None
        if i in order:
            order.remove(i)
            totals = retire(i)          # (taken before the slot's next run overwrites its result block)
        # the sweep kernel is bracketed by HIP events on every EVENT_EVERY-th step of the timed region (the two marker packets
        # cost a step 8 us, 5 % of it: profiles/r03/host_breakdown_headline.txt); its average duration over those launches is what
        # roofline.achieved divides by, and rocprofv3's kernel trace of the same command must agree (profiles/)
        sampled_slot[i] = timed and k % EVENT_EVERY == 0
        ctxs[i].set_option("profiling", 2 if sampled_slot[i] else 0)
        if depth > 1:
            batches[i].run_async(params)
            order.append(i)
            exchange(totals)            # ... and exchanged while both slots are running
        else:
            batches[0].run(params)
            exchange(retire(0))

    def drain():
        while order:
            exchange(retire(order.pop(0)))
        while pending:
            last_result[0] = reducer.collect(pending.pop(0))
        return last_result[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, timed=True)
    last = drain()
    sync()
    elapsed = time.perf_counter() - t0
    fill_ms, launches, timed_steps = acc["fill_ms"], acc["launches"], acc["timed_steps"]
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # ---- outside the timed region -----------------------------------------------------------------------
    # (a) the reduce on its own: blocking exchanges of this shard's totals
    reduce_ms = None
    if reducer is not None:
        tot = batch.ref_totals()
        for _ in range(3):
            reducer(tot, gids)
        sync()
        r0 = time.perf_counter()
        for _ in range(20):
            reducer(tot, gids)
        reduce_ms = (time.perf_counter() - r0) / 20 * 1e3
    # (b) a second, labelled figure: the step INCLUDING what OptAlignments hands back -- record index + every string
    mat_steps = max(3, min(args.steps, 20))
    ctx.set_option("profiling", 0)
    sync()
    m0 = time.perf_counter()
    for _ in range(mat_steps):
        batch.run(params)
        n_aln_all, n_chars = batch.materialise_all()
    ms_mat = (time.perf_counter() - m0) / mat_steps * 1e3
    # (c) a third: the same steps without any HIP events (the timed steps carry the two around the sweep that the roofline
    # figure needs) -- option "profiling" off, the library's default: what a caller that does not time the kernels pays
    ctx.set_option("profiling", 0)
    for _ in range(3):
        batch.run(params)
    sync()
    e0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run(params)
    sync()
    ms_noev = (time.perf_counter() - e0) / args.steps * 1e3
    # (d) the traceback kernel's duration: the same steps with every stage bracketed by events, outside the timed region
    ctx.set_option("profiling", 1)
    tb_steps = max(5, min(args.steps, 20))
    tb_ms = d2h_ms = 0.0
    solo_fill_ms, solo_launches = 0.0, 0
    for _ in range(tb_steps):
        batch.run(params)
        t = batch.timing()
        tb_ms += t.traceback_ms; d2h_ms += t.d2h_ms
        solo_fill_ms += t.fill_ms; solo_launches += t.fill_launches
    tb_ms *= args.steps / tb_steps; d2h_ms *= args.steps / tb_steps        # (reported per step below, like the sweep's)
    ctx.set_option("profiling", 2)
    gpu_scores, gpu_naln = batch.pair_results()
    same = all(np.array_equal(gpu_scores, b.pair_results()[0]) and np.array_equal(gpu_naln, b.pair_results()[1]) for b in batches[1:])
    winner = int(np.argmax(batch.ref_totals()))

    if rank == 0:
        total_cells = cells_rank * world * args.steps
        gcups = total_cells / elapsed / 1e9
        fill_avg_s = fill_ms / max(launches, 1) * 1e-3
        achieved = bytes_rank / fill_avg_s / 1e9 if fill_avg_s > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        prof = {}
        if os.path.exists(tf):
            try:
                prof = json.load(open(tf))
                traffic = None if tfused else prof.get("fill_kernel_hbm_bytes_per_launch")
            except Exception:
                prof, traffic = {}, None
        # supplementary: the bound that actually holds for this integer recurrence is instruction issue (DESIGN.md
        # 4.1).  Instructions per pair come from the committed PMC run and are only valid for the default workload.
        issue = None
        try:
            ipp = prof.get("sweep_insts_per_pair_headline")
            if ipp and (m, args.ref_len, len(refs), len(reads)) == (150, 2000, 1000, 1) and mode == 1 and fill_avg_s > 0 and not tfused:
                insts = (ipp["valu"] + ipp["salu"]) * len(refs) * len(reads)
                nominal = 1024 * 2.4e9 / 2.0         # SIMDs x max clock / 2 cycles per wave64 VALU (the guide's nominal rate)
                lone = 1024 * 2.4e9 / 4.0            # ... / 4 cycles: what ONE wave per SIMD can issue (guide, 'one wave alone: 4')
                issue = {"insts_per_launch": insts, "insts_source": ipp.get("source"),
                         "achieved_ginst_s": round(insts / fill_avg_s / 1e9, 1),
                         "frac_of_nominal_2cyc": round(insts / fill_avg_s / nominal, 4),
                         "frac_of_one_wave_per_simd_4cyc": round(insts / fill_avg_s / lone, 4)}
        except Exception:
            issue = None
        out = {
            "metric": "GCUPS", "value": round(gcups, 3), "unit": "GCUPS (1e9 DP cell updates/s, full path)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "alignments_per_s": round(len(refs) * len(reads) * world * args.steps / elapsed, 1),
            "in_flight": depth,
            "ms_per_step_one_in_flight": round(ms_noev, 4),
            "gcups_one_in_flight": round(cells_rank / (ms_noev * 1e-3) / 1e9, 3),
            "ms_per_step_materialised": round(ms_mat, 4),
            "ms_per_step_without_events": round(ms_noev, 4),
            "materialised": {"what": "run (both aligned strings of every alignment are written by the traceback kernels) + the index over "
                                     "every alignment (swmi_batch_materialise_all): everything OptAlignments returns; rank 0, no events, "
                                     "one step after the other, outside the timed region", "steps": mat_steps,
                             "alignments": int(n_aln_all), "chars": int(n_chars),
                             "gcups": round(cells_rank / (ms_mat * 1e-3) / 1e9, 3)},
            "config": {"workload": "configs[1]: 1 read x %d bp vs %d refs x %d bp per GPU, scores 5/-3/-4, mode %d: %s"
                                   % (m, len(refs), args.ref_len, mode, "transposed sweep (column checkpoints) + block re-sweeps + walks + result records, ONE kernel (option tfused)" if tfused else WORKLOAD_OF_MODE[mode]),
                       "pairs_per_gpu": len(refs) * len(reads), "cells_per_step_per_gpu": cells_rank,
                       "steps_in_flight": "%d per GPU: each step is one full pass over one resident batch (own context, stream, workspace and "
                                          "pinned result block); with 2, step k+1's sweep runs in the tail of step k's traceback launch. "
                                          "ms_per_step_one_in_flight / gcups_one_in_flight: the same steps strictly one after the other" % depth
                                          if depth > 1 else "1: strictly one step after the other",
                       "parallelism": "references sharded over %d rank(s); max/top-K reduce %s"
                                      % (world, ("over gloo (one-GPU rehearsal)" if one_gpu else "over RCCL") if world > 1 else "local")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": kernel_name, "kernel_avg_ms": round(fill_avg_s * 1e3, 4),
                         "alg_bytes_per_launch": bytes_rank,
                         "kernel_gcups": round(cells_rank / fill_avg_s / 1e9, 2) if fill_avg_s > 0 else None,
                         "traceback_avg_ms": round(tb_ms / args.steps, 4), "d2h_avg_ms": round(d2h_ms / args.steps, 4),
                         "kernel_timed_launches": launches,
                         "kernel_timing": "HIP events on the launching context's stream around the sweep kernel of every %d-th timed step (%d of %d steps)%s" % (
                             EVENT_EVERY, timed_steps, args.steps, "; with two steps in flight the sweep shares the chip with the other step's traceback" if depth > 1 else ""),
                         "one_in_flight": {"kernel_avg_ms": round(solo_fill_ms / max(solo_launches, 1), 4),
                                           "frac": round(bytes_rank / (solo_fill_ms / max(solo_launches, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if solo_fill_ms > 0 else None,
                                           "what": "the same kernel's duration with one step after the other (the loop that times the traceback)"},
                         "traceback_note": "measured in %d extra steps with every stage bracketed by events, outside the timed region "
                                           "(the timed steps bracket the sweep only)" % tb_steps},
            "check": {"winner_ref": winner, "winner_total": batch.ref_total(winner), "in_flight_batches_identical": bool(same)},
        }
        if reduce_ms is not None:
            out["reduce_ms"] = round(reduce_ms, 4)
            out["reduce"] = {"what": "one blocking max-with-ties exchange (all_gather of {local max, winners}), measured "
                                     "on its own outside the timed region; inside it the exchange overlaps the next shard",
                             "result": [int(last[0]), [int(x) for x in last[1]][:8]] if last else None}
        if issue:
            out["roofline"]["issue"] = issue
        if not args.no_cpu_baseline and world == 1:
            base, chk = cpu_baseline(refs, reads, gpu_scores, gpu_naln)
            out["cpu_baseline"] = base
            out["check"].update(chk)
        print(json.dumps(out), flush=True)
    for b in batches:
        b.free()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
    if world > 1 or rehearse_rccl:
        dist.destroy_process_group()


def strong_scaling(args, world, rank, local_rank, one_gpu, dev):
    """BASELINE.json configs[3]: ONE reference set (NCBI-shaped, --n-refs) x --n-reads reads, the references sharded by length
    over the ranks (distributed.shard_by_length), every rank streaming its shard through its GPU in chunks of --chunk-kb KiB
    of references (each chunk = its references x all reads, full path: sweep + tied maxima + traceback + both strings of every
    alignment written to host memory), then the driver's reduce over RCCL: max total with ties (Distribution.java:341-353,
    600-613) + top-K, and the winners aligned once more into results that are kept (a chunk's records are dropped once its
    totals are taken, like the reference's driver drops every non-winner's alignments).  A step = all of that once."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import sparksmithwaterman_amd as sw
    from sparksmithwaterman_amd import synth
    from sparksmithwaterman_amd import distributed as swd

    g0 = time.perf_counter()
    refs, reads = synth.config_multi_read(args.n_refs, args.n_reads, seed=3)       # (every rank the same set: deterministic)
    gen_s = time.perf_counter() - g0
    lengths = np.fromiter((len(r) for r in refs), dtype=np.int64, count=len(refs))
    local_ids = swd.shard_by_length(lengths, rank, world)
    mine = [refs[int(i)] for i in local_ids]
    read_lens, read_cnt = np.unique([len(q) for q in reads], return_counts=True)
    cells_rank = int(lengths[local_ids].sum()) * int(sum(len(q) for q in reads))
    cells_all = int(lengths.sum()) * int(sum(len(q) for q in reads))
    bytes_rank = sum(int(c) * alg_bytes(int(mq), int(n)) for n in lengths[local_ids] for mq, c in zip(read_lens, read_cnt))

    ctx = sw.Context(local_rank)
    ctx.set_option("profiling", 1)          # (every stage: a chunk runs for tens of milliseconds, the marker packets do not matter)
    ctx.set_option("stream_keep_records", 0)
    for kv in args.opt:
        name, _, value = kv.partition("=")
        ctx.set_option(name, int(value))
    params = sw.make_params()
    rdev = None if (one_gpu or world == 1) else dev
    mine_ids = set(int(x) for x in local_ids)
    last = {}

    def step():
        st = ctx.stream(reads, params, slots=3, chunk_bytes=args.chunk_kb << 10)
        try:
            t0 = time.perf_counter()
            st.push(mine).finish()
            totals = st.totals().copy()
            stats = st.stats()
            t1 = time.perf_counter()
            best, winners = swd.global_max_with_ties(totals, local_ids, device=rdev)
            top = swd.global_top_k(totals, local_ids, args.top_k, device=rdev)
            t2 = time.perf_counter()
            # the winners this rank owns, aligned again into results that stay (MapRef's output for them)
            own = [w for w in winners if w in mine_ids][:64]
            sites = 0
            if own:
                wb = ctx.upload([refs[w] for w in own], reads).run(params)
                sites = sum(len(s_) + d_ for _, d_, s_ in wb.ref_sites_packed())
                wb.free()
            t3 = time.perf_counter()
        finally:
            st.close()
        last.update(totals=totals, best=best, winners=winners, top=top, sweep_ms=stats.gpu_sweep_ms, tb_ms=stats.gpu_traceback_ms,
                    chunks=stats.chunks, align_s=t1 - t0, reduce_s=t2 - t1, winners_s=t3 - t2, winner_sites=sites)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    sweep_ms = tb_ms = align_s = reduce_s = winners_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        sweep_ms += last["sweep_ms"]; tb_ms += last["tb_ms"]
        align_s += last["align_s"]; reduce_s += last["reduce_s"]; winners_s += last["winners_s"]
    sync()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # a labelled second figure, outside the timed region: the driver's shortcut.  The reference's driver keeps only the winning
    # references' alignments (Distribution.java:341-353), so a scores-only pass over the shard (option scores_only: the sweep
    # kernels alone) + the full path for the winners (the "winners" part of every step above) gives the same output.
    full_totals = last["totals"].copy()
    ctx.set_option("scores_only", 1)
    sync()
    s0 = time.perf_counter()
    st = ctx.stream(reads, params, slots=3, chunk_bytes=args.chunk_kb << 10)
    st.push(mine).finish()
    so_totals = st.totals().copy()
    st.close()
    torch.cuda.synchronize()
    scores_only_s = time.perf_counter() - s0
    ctx.set_option("scores_only", 0)
    so_equal = bool((so_totals == full_totals).all())

    # the reduce on its own (blocking exchanges of this shard's totals), outside the timed region
    for _ in range(2):
        swd.global_max_with_ties(last["totals"], local_ids, device=rdev)
    sync()
    r0 = time.perf_counter()
    for _ in range(10):
        swd.global_max_with_ties(last["totals"], local_ids, device=rdev)
        swd.global_top_k(last["totals"], local_ids, args.top_k, device=rdev)
    reduce_ms = (time.perf_counter() - r0) / 10 * 1e3

    if rank == 0:
        sweep_s = sweep_ms / args.steps * 1e-3
        achieved = bytes_rank / sweep_s / 1e9 if sweep_s > 0 else 0.0
        out = {
            "metric": "GCUPS", "value": round(cells_all * args.steps / elapsed / 1e9, 3), "unit": "GCUPS (1e9 DP cell updates/s, full path)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "alignments_per_s": round(len(refs) * len(reads) * args.steps / elapsed, 1),
            "reduce_ms": round(reduce_ms, 4),
            "config": {"workload": "configs[3]: %d reads x 150 bp vs ONE set of %d NCBI-shaped references (median 1,609 bp), sharded by length over "
                                   "%d rank(s), streamed in chunks of %d KiB of references; scores 5/-3/-4; full path for every pair (sweep + tied "
                                   "maxima + traceback + both strings of every alignment to host memory), totals -> max-with-ties + top-%d over %s, "
                                   "winners aligned again into kept results" % (len(reads), len(refs), world, args.chunk_kb, args.top_k,
                                                                            ("gloo (one-GPU rehearsal)" if one_gpu else "RCCL") if world > 1 else "one rank"),
                       "pairs": len(refs) * len(reads), "pairs_rank0": len(mine) * len(reads), "cells": cells_all, "cells_rank0": cells_rank,
                       "parallelism": "references sharded by length over %d rank(s); no data-path collective" % world},
            "rank0": {"align_s_per_step": round(align_s / args.steps, 4), "reduce_s_per_step": round(reduce_s / args.steps, 5),
                      "winners_s_per_step": round(winners_s / args.steps, 4), "chunks": int(last["chunks"]),
                      "generate_inputs_s": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "sw_sweep_winmax_kernel (sum over rank 0's chunks, three streams)",
                         "kernel_ms_per_step": round(sweep_ms / args.steps, 3), "alg_bytes_per_step_rank0": bytes_rank,
                         "traceback_ms_per_step": round(tb_ms / args.steps, 3)},
            "scores_only_pass": {"what": "rank 0's shard once more with option scores_only (sweep kernels only), outside the timed region: "
                                         "what a driver that aligns only its winners in full would run first",
                                 "seconds": round(scores_only_s, 4), "gcups_rank0": round(cells_rank / scores_only_s / 1e9, 1),
                                 "totals_equal_full_path": so_equal},
            "reduce": {"best_total": int(last["best"]), "winners": [int(w) for w in last["winners"]][:8],
                       "top_k": [[int(a), int(b_)] for a, b_ in last["top"]], "winner_sites_rank0": int(last["winner_sites"])},
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import sw_oracle as orc
            cores = effective_cores()
            # bounded sample: the first references of the set x all reads, ~15 s of CPU work; totals compared with the GPU's
            per_ref = float(np.mean(lengths[:64])) * sum(len(q) for q in reads)
            n_s = int(max(8, min(len(refs), 3.0e9 * cores / 16 * 15 / per_ref)))
            r = orc.bench(refs[:n_s], reads, nthreads=cores, per_pair=True)
            want = np.asarray(r["pair_score"], dtype=np.int64).reshape(n_s, len(reads)).sum(axis=1)
            pos = {int(g): k for k, g in enumerate(local_ids)}
            got = np.array([int(last["totals"][pos[g]]) for g in range(n_s)], dtype=np.int64)
            gc = r["cells"] / r["seconds"] / 1e9
            out["cpu_baseline"] = {"value": round(gc, 4), "unit": "GCUPS", "cores": cores, "kind": "port",
                                   "sample": "the first %d references x all %d reads (%.1f s); C restatement of SmithWaterman.OptAlignments, "
                                             "persistent pool of %d pthreads" % (n_s, len(reads), r["seconds"], cores)}
            out["check"] = {"refs_compared": n_s, "mismatches": int((want != got).sum()),
                            "compared": "per-reference totals over all reads, GPU vs oracle, same run"}
        print(json.dumps(out), flush=True)
    ctx.close()


def cpu_baseline(refs, reads, gpu_scores, gpu_naln):
    """The oracle's full path (C restatement of the Java code, NOT the JVM) on the host cores this job owns, on a
    bounded sample: the same 1k-reference batch, `reps` passes (~12 s) by a thread pool that exists before the clock
    starts.  The same pass yields every pair's score and alignment count, compared here with the GPU's."""
    from oracle import sw_oracle as orc
    cores = effective_cores()
    r = orc.bench(refs[:max(4 * cores, 64)], reads, nthreads=cores)            # calibration
    rate = r["cells"] / max(r["seconds"], 1e-9)
    per_pass = sum(len(x) for x in refs) * sum(len(q) for q in reads)
    reps = int(max(1, min(200, rate * 12.0 / per_pass)))
    while reps * len(refs) * len(reads) < 16 * cores:                          # >= 16 pairs per thread
        reps += 1
    r = orc.bench(refs, reads, nthreads=cores, reps=reps, per_pair=True)
    mism = sum(1 for k in range(len(refs) * len(reads))
               if int(gpu_scores[k]) != r["pair_score"][k] or int(gpu_naln[k]) != r["pair_naln"][k])
    gc = r["cells"] / r["seconds"] / 1e9
    base = {"value": round(gc, 4), "unit": "GCUPS", "cores": cores, "kind": "port",
            "mcups_per_core": round(gc * 1e3 / cores, 1),
            "sample": "%d passes over the same %d pairs (%.1f s); C restatement of SmithWaterman.OptAlignments "
                      "(full int + char matrices, all tied maxima, stack traceback), one pair per task, persistent "
                      "pool of %d pthreads" % (reps, len(refs) * len(reads), r["seconds"], cores)}
    chk = {"pairs_compared": len(refs) * len(reads), "mismatches": mism,
           "compared": "score and number of alignments of every pair, GPU vs oracle, same run",
           "sum_score": [int(sum(int(x) for x in gpu_scores)), r["sum_score"]],
           "sum_alignments": [int(sum(int(x) for x in gpu_naln)), r["sum_aln"]]}
    return base, chk


if __name__ == "__main__":
    main()
