#!/usr/bin/env python3
"""Throughput of the headline step with one, two and three batches in flight (one context + stream + host thread each,
swmi_batch_run_async): does the next batch's sweep fill the tail of the previous batch's traceback launch?  No oracle."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparksmithwaterman_amd as sw      # noqa: E402
from sparksmithwaterman_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
refs, reads = synth.config_1k(n_refs, 2000, 150, seed=1)
cells = sum(len(r) for r in refs) * len(reads[0])
STEPS = 200
for depth in (1, 2, 3, 1, 2):
    ctxs = [sw.Context(0) for _ in range(depth)]
    bs = [c.upload(refs, reads) for c in ctxs]
    for b in bs:
        b.run(); b.run()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        if depth == 1:
            for _ in range(STEPS):
                bs[0].run()
        else:
            fl = [False] * depth
            for k in range(STEPS):
                i = k % depth
                if fl[i]:
                    bs[i].wait()
                bs[i].run_async(); fl[i] = True
            for i in range(depth):
                if fl[i]:
                    bs[i].wait()
        dt = (time.perf_counter() - t0) / STEPS
        best = min(best, dt)
    sc = [int(x) for x in bs[-1].pair_results()[0][:4]]
    print("%d in flight: %.4f ms per step, %.0f GCUPS   (scores of the first pairs %s)" % (depth, best * 1e3, cells / best / 1e9, sc), flush=True)
    for b in bs:
        b.free()
    for c in ctxs:
        c.close()
