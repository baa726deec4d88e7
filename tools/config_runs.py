#!/usr/bin/env python3
"""BASELINE.json configs[2], [3] (one GPU's share) and [4] on one MI355X through the product path, each CHECKED against the
oracle -- not run-to-run -- and each with its HBM-roofline fraction (algorithmic bytes over the sweep
kernels' time).  Prints a markdown table (profiles/r03/config_runs.md).

    python tools/config_runs.py [--c2-refs 1000000] [--c3 12500x1000] [--c4 8x8x10000] > gpurun_out/r03/config_runs.md

configs[2]  NCBI-shaped references x one 150 bp read, FASTA file -> results, streamed (swmi_stream_push_file).  Check: the sum
            of all scores, the sum of all alignment counts and the winner against one oracle pass over every pair.
configs[3]  ONE GPU's share of "10 k reads x 100 k references on 8 GPUs": 12,500 NCBI-shaped references x R reads (default
            1,000; 10,000 is the full share), streamed in chunks of a few hundred references (every chunk is refs x reads
            pairs).  Check: per-reference totals of a sample of >= 200 references against the oracle's pass over those
            references x all reads, and MapRef's sorted match sites (every string) of 24 of them.
configs[4]  10 kbp x 10 kbp pairs as ONE batch of 64 pairs (8 references x 8 reads; read k is reference k with 10 %
            substitutions + 2 % indels, the other 56 pairs are unrelated).  Check: score, number of alignments and every
            alignment string of >= 4 pairs (2 related, 2 unrelated) against the oracle's full matrices.
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

HBM_PEAK = 8000.0e9


def alg_bytes(m, n):
    return -(-n // 4) + -(-m // 4) + -(-(m * n) // 4) + -(-(m + n) // 4) + 16


def cores():
    c = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            c = max(1, min(c, -(-int(quota) // int(period))))
    except Exception:
        pass
    return c


def row(name, pairs, cells, wall_s, sweep_s, abytes, aln, mism, note):
    frac = abytes / sweep_s / HBM_PEAK if sweep_s > 0 else float("nan")
    print("| %s | %d | %.3e | %.3f | %.1f | %.4f | %.1f | %.4f | %d | mismatches: %d (%s) |" % (
        name, pairs, cells, wall_s, cells / wall_s / 1e9, sweep_s, cells / sweep_s / 1e9 if sweep_s > 0 else float("nan"), frac, aln, mism, note))
    sys.stdout.flush()


def config2(ctx, sw, synth, orc, n_refs, tmpdir):
    from stream_run import write_fasta
    refs, reads = synth.config_ncbi(n_refs, read_len=150, seed=2)
    path = os.path.join(tmpdir, "swmi_c2_%d.fa" % n_refs)
    write_fasta(path, refs)
    m = len(reads[0])
    cells = sum(len(r) for r in refs) * m
    ab = sum(alg_bytes(m, len(r)) for r in refs)
    best = None
    try:
        for _ in range(2):
            st = ctx.stream(reads, slots=3, chunk_bytes=32 << 20)
            t0 = time.perf_counter()
            st.push_file(path, ">gi", 6)
            st.finish()
            wall = time.perf_counter() - t0
            totals = st.totals().astype(np.int64)
            n_aln = sum(int(b.pair_results()[1].sum()) for _, b in st.chunks())
            s = st.stats()
            st.close()
            if best is None or wall < best[0]:
                best = (wall, s.gpu_sweep_ms * 1e-3, totals, n_aln)
    finally:
        os.unlink(path)
    wall, sweep_s, totals, n_aln = best
    ob = orc.bench(refs, reads, nthreads=cores(), per_pair=True)
    mism = int((np.asarray(ob["pair_score"], dtype=np.int64) != totals).sum())
    mism += int(ob["sum_aln"] != n_aln)
    row("configs[2]: %d NCBI-shaped refs x 1 read (150 bp), FASTA file -> results, streamed" % n_refs, n_refs, cells, wall, sweep_s, ab,
        n_aln, mism, "every pair's score and the sum of alignment counts vs one oracle pass over all %d pairs, %.0f s on %d threads"
        % (n_refs, ob["seconds"], cores()))


def config3(ctx, sw, synth, orc, n_refs, n_reads, chunk_kb, sample, sites):
    refs, reads = synth.config_multi_read(n_refs, n_reads, seed=3)
    cells = sum(len(r) for r in refs) * sum(len(q) for q in reads)
    lens_q = np.array([len(q) for q in reads])
    uq, cq = np.unique(lens_q, return_counts=True)
    ab = sum(int(c) * alg_bytes(int(mq), len(r)) for r in refs for mq, c in zip(uq, cq))
    st = ctx.stream(reads, slots=3, chunk_bytes=chunk_kb << 10)
    t0 = time.perf_counter()
    st.push(refs)
    st.finish()
    wall = time.perf_counter() - t0
    totals = st.totals().astype(np.int64)
    s = st.stats()
    chunks = st.chunks()
    n_aln = sum(int(b.pair_results()[1].sum()) for _, b in chunks)
    # ---- checks against the oracle on a sample of references ----
    rng = np.random.default_rng(12345)
    pick = np.sort(rng.choice(n_refs, size=min(sample, n_refs), replace=False))
    ob = orc.bench([refs[i] for i in pick], reads, nthreads=cores(), per_pair=True)
    want = np.asarray(ob["pair_score"], dtype=np.int64).reshape(len(pick), n_reads)
    mism = int((want.sum(axis=1) != totals[pick]).sum())
    firsts = [f for f, _ in chunks]

    def view_of(r):
        k = int(np.searchsorted(firsts, r, side="right")) - 1
        return chunks[k][1], r - chunks[k][0]

    for x, r in enumerate(pick):                          # every pair score and alignment count of the sample
        b, loc = view_of(int(r))
        sc, na = b.pair_results()
        mism += int((sc[loc * n_reads:(loc + 1) * n_reads].astype(np.int64) != want[x]).sum())
        mism += int((na[loc * n_reads:(loc + 1) * n_reads].astype(np.int64) != np.asarray(ob["pair_naln"], dtype=np.int64).reshape(len(pick), n_reads)[x]).sum())
    site_refs = [int(r) for r in pick[:sites]]
    with ThreadPoolExecutor(max_workers=cores()) as ex:    # (ctypes releases the GIL inside the oracle)
        wants = list(ex.map(lambda r: orc.map_ref((">gi|ref%d" % r, refs[r]), reads), site_refs))
    n_sites = 0
    for r, (wt, (_, ws)) in zip(site_refs, wants):
        b, loc = view_of(r)
        got = b.ref_match_sites(loc)
        n_sites += len(ws)
        mism += int(b.ref_total(loc) != wt) + int(got != ws)
    st.close()
    row("configs[3], one GPU's share: %d NCBI-shaped refs x %d reads (150 bp), streamed in chunks of %d KiB of references" % (n_refs, n_reads, chunk_kb),
        n_refs * n_reads, cells, wall, s.gpu_sweep_ms * 1e-3, ab, n_aln, mism,
        "totals, every pair's score and alignment count of %d sampled references (%d pairs) vs the oracle; MapRef's sorted match sites "
        "-- %d sites, every string -- of %d of them" % (len(pick), len(pick) * n_reads, n_sites, len(site_refs)))


def config4(ctx, sw, synth, orc, n_refs, n_reads, length):
    refs, reads = synth.config_long(max(n_refs, n_reads), length, seed=4)
    refs, reads = refs[:n_refs], reads[:n_reads]
    cells = sum(len(r) for r in refs) * sum(len(q) for q in reads)
    ab = sum(alg_bytes(len(q), len(r)) for r in refs for q in reads)
    ctx.set_option("profiling", 1)
    b = ctx.upload(refs, reads)
    b.run()
    best, sweep_s = None, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        b.run()
        w = time.perf_counter() - t0
        if best is None or w < best:
            best, sweep_s = w, b.timing().fill_ms * 1e-3
    sc, na = b.pair_results()
    mism = 0
    checked = [(0, 0), (n_refs - 1, n_reads - 1), (0, n_reads - 1), (n_refs - 1, 0)]
    for r, q in checked:
        es, ea = orc.opt_alignments((refs[r], reads[q]))
        pair = r * n_reads + q
        mism += int(b.score(pair) != es) + int(b.alignments(pair) != ea)
    t = b.timing()
    row("configs[4]: ONE batch of %d pairs of %d x %d bp (%d refs x %d reads), strip pipeline (%d strip fallbacks)" % (
        n_refs * n_reads, length, length, n_refs, n_reads, t.strip_fallbacks), n_refs * n_reads, cells, best, sweep_s, ab, int(na.sum()), mism,
        "score, alignment count and every alignment string of %d pairs (2 related, 2 unrelated) vs the oracle's full matrices" % len(checked))
    b.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c2-refs", type=int, default=1000000)
    ap.add_argument("--c3", default="12500x1000", help="refs x reads of ONE GPU's share of configs[3]")
    ap.add_argument("--c3-chunk-kb", type=int, default=512)
    ap.add_argument("--c3-sample", type=int, default=200)
    ap.add_argument("--c3-sites", type=int, default=24)
    ap.add_argument("--c4", default="8x8x10000")
    ap.add_argument("--only", default="2,3,4")
    ap.add_argument("--tmp", default="/dev/shm")
    args = ap.parse_args()
    import sparksmithwaterman_amd as sw
    from sparksmithwaterman_amd import synth
    from oracle import sw_oracle as orc
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ctx = sw.Context(0)
    ctx.set_option("profiling", 1)
    print("| config | pairs | cells | wall s (inputs -> results) | GCUPS (full path) | sweep kernels s | GCUPS (sweep) | HBM roofline fraction (algorithmic bytes / sweep time / 8 TB/s) | alignments | check |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    sys.stdout.flush()
    only = set(args.only.split(","))
    if "2" in only:
        config2(ctx, sw, synth, orc, args.c2_refs, args.tmp)
    if "3" in only:
        nr, nq = (int(x) for x in args.c3.split("x"))
        config3(ctx, sw, synth, orc, nr, nq, args.c3_chunk_kb, args.c3_sample, args.c3_sites)
    if "4" in only:
        a, b_, ln = (int(x) for x in args.c4.split("x"))
        config4(ctx, sw, synth, orc, a, b_, ln)
    ctx.close()


if __name__ == "__main__":
    main()
