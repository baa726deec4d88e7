#!/usr/bin/env python3
"""What the three HIP events per run (option "profiling") cost at the headline config: ms per run with and without them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sparksmithwaterman_amd as sw
from sparksmithwaterman_amd import synth
refs, reads = synth.config_1k()
ctx = sw.Context(0)
b = ctx.upload(refs, reads)
p = sw.make_params()
for rep in range(3):
    for prof in (1, 0):
        ctx.set_option("profiling", prof)
        for _ in range(20):
            b.run(p)
        t0 = time.perf_counter()
        for _ in range(300):
            b.run(p)
        print("profiling %d: %.4f ms per run" % (prof, (time.perf_counter() - t0) / 300 * 1e3), flush=True)
