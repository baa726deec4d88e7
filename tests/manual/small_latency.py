#!/usr/bin/env python3
"""The per-call floor of a tiny batch (one pair of 80 x 400, EngineerData's smallest point): wall per run and the kernels'
share of it, for the automatic pipeline, the fused traceback and the resident kernel (no oracle)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sparksmithwaterman_amd as sw
REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"
READ_80 = "AATTTTAGTCTCTCCCTACCCTTTTGGACAGAGCTTCCTGTCCTCTCATTTCACAGGTTATGCAACAGAGGGTTCTGTGT"
ctx = sw.Context(0)
for opts in ({}, {"profiling": 1}, {"tb_split": 0}, {"resident": 1}):
    for k, v in opts.items():
        ctx.set_option(k, v)
    b = ctx.upload([REF * 5], [READ_80])
    for _ in range(20):
        b.run()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); b.run(); ts.append(time.perf_counter() - t0)
    ts.sort()
    t = b.timing()
    print(opts, "median %.1f us  min %.1f us   fill %.1f us traceback %.1f us" % (ts[100] * 1e6, ts[0] * 1e6, t.fill_ms * 1e3, t.traceback_ms * 1e3), flush=True)
    b.free()
    for k in opts:
        ctx.set_option(k, {"profiling": 0, "tb_split": -1, "resident": -1}[k])
ctx.close()
