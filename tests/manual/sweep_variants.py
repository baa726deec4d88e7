#!/usr/bin/env python3
"""The sweep of 1000 pairs x 2000 bp for a byte alphabet (generic cell stream) and for reads of 60 / 250 rows (R = 1 / 4): wall
and sweep-kernel time per run, one step at a time (A/B of kernel builds through SWMI_LIB; no oracle)."""
import os, sys, time, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sparksmithwaterman_amd as sw
rng = random.Random(3)
for alpha, m in (("ACGTKMSW", 150), ("ACGTKMSW", 250), ("ACGT", 250), ("ACGT", 60)):
    refs = ["".join(rng.choice(alpha) for _ in range(2000)) for _ in range(1000)]
    read = refs[0][100:100 + m]
    ctx = sw.Context(0); ctx.set_option("profiling", 1)
    b = ctx.upload(refs, [read])
    for _ in range(5): b.run()
    f = 0.0; t0 = time.perf_counter()
    for _ in range(30):
        b.run(); f += b.timing().fill_ms
    dt = (time.perf_counter() - t0) / 30
    print("%-9s m=%-3d  %.4f ms per run, sweep %.4f ms" % (alpha, m, dt * 1e3, f / 30), flush=True)
    b.free(); ctx.close()
