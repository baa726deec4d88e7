#!/usr/bin/env python3
"""EngineerData-shaped one-factor sweeps: the reference's own benchmark shapes
(src/metrics/EngineerData.java:51-224, src/metrics/ExecutionTimesReference.java:43-124) on the GPU path, with the
CPU oracle timed beside it on the small points.  References are REF repeated (EngineerData.java:118), so tied
maxima -- and therefore the multi-alignment output path -- are the norm, exactly as in the reference's data.

    python tests/manual/sweep_bench.py [--quick] > gpurun_out/sweeps.md
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"       # EngineerData.java:23
READ_80 = "AATTTTAGTCTCTCCCTACCCTTTTGGACAGAGCTTCCTGTCCTCTCATTTCACAGGTTATGCAACAGAGGGTTCTGTGT"   # :26
READ_20 = "ACTGACTGACTGACTGACTG"                                                                  # :29


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--mode", type=int, default=-1, help="pipeline: -1 automatic (default), 0, 1, 2")
    ap.add_argument("--col-chunks", type=int, default=0, help="0 automatic, 1 never, N force")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="any swmi_set_option knob (A/B runs), e.g. device_strings=0")
    ap.add_argument("--only", default="1,2,3,4", help="which of the four sweeps to run")
    args = ap.parse_args()
    import sparksmithwaterman_amd as sw
    from oracle import sw_oracle as orc

    ctx = sw.Context(0)
    ctx.set_option("mode", args.mode)
    ctx.set_option("col_chunks", args.col_chunks)
    for kv in args.opt:
        name, _, value = kv.partition("=")
        ctx.set_option(name, int(value))
    print("pipeline option: mode %d, col_chunks %d%s" % (args.mode, args.col_chunks, "".join(", " + kv for kv in args.opt)))
    only = set(args.only.split(","))
    modes = []

    def run(refs, reads, check):
        b = ctx.upload(refs, reads)
        b.run()                                   # warm-up (and, in automatic mode, the sampled pipeline choice)
        dt = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            b.run()
            dt = min(dt, time.perf_counter() - t0)
        modes.append((b.pipeline_mode(), b.timing().col_chunks))
        cells = sum(map(len, refs)) * sum(map(len, reads))
        n_aln = sum(b.n_alignments(p)[0] for p in range(len(refs) * len(reads)))
        cpu = None
        if check:
            r = orc.bench(refs, reads, nthreads=min(16, os.cpu_count() or 1))
            assert r["sum_score"] == sum(b.score(p) for p in range(len(refs) * len(reads)))
            assert r["sum_aln"] == n_aln
            cpu = r["cells"] / r["seconds"] / 1e9
        b.free()
        return cells, dt, n_aln, cpu

    def table(title, rows):
        print("\n### %s\n" % title)
        print("| point | cells | alignments | GPU ms | GPU GCUPS | pipeline (mode, column chunks) | CPU oracle GCUPS (all cores) |")
        print("|---|---|---|---|---|---|---|")
        for (name, (cells, dt, n_aln, cpu)), md in zip(rows, modes[-len(rows):]):
            print("| %s | %.3g | %d | %.3f | %.1f | %d, %d | %s |" % (name, cells, n_aln, dt * 1e3, cells / dt / 1e9, md[0], md[1],
                                                                  "%.2f" % cpu if cpu else "-"))

    q = args.quick
    # test 1: number of reads (80 bp) against one 400 bp reference      EngineerData.java:51-79
    pts = [20, 50, 100, 200] if q else [20, 50, 100, 200, 400, 800, 1600]
    if "1" in only:
      table("reads sweep: N reads x 80 bp vs 1 ref x 400 bp", [(str(n), run([REF * 5], [READ_80] * n, n <= 200)) for n in pts])
    # test 2: read length, 5 reads                                       EngineerData.java:87-104
    pts = [20, 100, 300] if q else [20, 40, 80, 100, 200, 300, 400, 500]
    if "2" in only:
      table("read-length sweep: 5 reads x L vs 1 ref x 4000 bp",
          [(str(L), run([REF * 50], [(READ_80 * 7)[:L]] * 5, L <= 200)) for L in pts])
    # test 3: number of references (400 bp), one 80 bp read              EngineerData.java:116-169
    pts = [1, 100, 1000] if q else [1, 10, 100, 1000, 4000, 10000, 40000]
    if "3" in only:
      table("#references sweep: 1 read x 80 bp vs N refs x 400 bp",
          [(str(n), run([REF * 5] * n, [READ_80], n <= 1000)) for n in pts])
    # test 4: reference length, one reference, one 80 bp read            EngineerData.java:178-224
    pts = [80, 1600, 16000] if q else [80, 400, 1600, 8000, 32000, 128000]
    if "4" in only:
      table("reference-length sweep: 1 read x 80 bp vs 1 ref x L",
          [(str(L), run([REF * (L // 80)], [READ_80], L <= 8000)) for L in pts])
    ctx.close()


if __name__ == "__main__":
    main()
