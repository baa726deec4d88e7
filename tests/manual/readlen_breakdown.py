#!/usr/bin/env python3
"""Where the read-length sweep's long points (EngineerData.java:87-104: 5 reads x L bp vs one 4000 bp periodic reference) spend
their time: sweep / traceback kernel times (HIP events) and wall, per read length."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparksmithwaterman_amd as sw      # noqa: E402

REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"
READ_80 = "AATTTTAGTCTCTCCCTACCCTTTTGGACAGAGCTTCCTGTCCTCTCATTTCACAGGTTATGCAACAGAGGGTTCTGTGT"
ctx = sw.Context(0)
ctx.set_option("profiling", 1)
for opt in sys.argv[1:]:
    k, _, v = opt.partition("=")
    ctx.set_option(k, int(v))
print("L     wall ms  sweep ms  traceback ms  alignments  mean path  chunk sweeps")
for L in (100, 200, 256, 257, 300, 400, 500):
    b = ctx.upload([REF * 50], [(READ_80 * 7)[:L]] * 5)
    b.run()
    best = (1e9, None)
    for _ in range(5):
        t0 = time.perf_counter()
        b.run()
        dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, b.timing())
    n_aln = sum(b.n_alignments(p)[0] for p in range(5))
    na, nc = b.materialise_all()
    print("%-5d %.3f    %.3f     %.3f         %d         %.0f        %d" % (L, best[0] * 1e3, best[1].fill_ms, best[1].traceback_ms, n_aln,
                                                                        nc / 2 / max(na, 1), best[1].col_chunks))
    b.free()
ctx.close()
