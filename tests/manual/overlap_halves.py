#!/usr/bin/env python3
"""can the idle half of the chip be used at the headline by overlapping INSIDE a step -- the 1000 pairs split in
two halves on two streams, the traceback of half A (4 waves per SIMD, latency-bound) running beside the sweep of half B (one
wave per SIMD)?  Two contexts (each its own HIP stream) on two host threads; per variant the wall time until BOTH halves'
results are in host memory, best of several repetitions of `iters` steps.

    python tests/manual/overlap_halves.py > gpurun_out/r03/overlap_halves.txt

  one batch       the product path: 1000 pairs, one sweep launch, one traceback launch
  back to back    half A then half B on ONE stream (what splitting costs by itself)
  side by side    both halves started together on two streams (their sweeps share the SIMDs)
  staggered       half B started `delay` us after half A, so that B's sweep runs beside A's traceback
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparksmithwaterman_amd as sw           # noqa: E402
from sparksmithwaterman_amd import synth      # noqa: E402


def main():
    iters, reps = 200, 5
    refs, reads = synth.config_1k(1000, 2000, 150, seed=1)
    params = sw.make_params()
    ca, cb = sw.Context(0), sw.Context(0)
    whole = ca.upload(refs, reads)
    ha, hb = ca.upload(refs[:500], reads), cb.upload(refs[500:], reads)
    for b in (whole, ha, hb):
        for _ in range(5):
            b.run(params)

    def best(fn):
        out = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            out = min(out, (time.perf_counter() - t0) / iters)
        return out * 1e3

    def one_batch():
        for _ in range(iters):
            whole.run(params)

    def back_to_back():
        hb2 = ca.upload(refs[500:], reads)
        for _ in range(3):
            hb2.run(params)
        t = [0.0]

        def f():
            for _ in range(iters):
                ha.run(params)
                hb2.run(params)
        r = best(f)
        hb2.free()
        return r

    def two_streams(delay_us):
        def f():
            start = threading.Barrier(2)
            go = [threading.Event() for _ in range(iters)]
            done_b = [threading.Event() for _ in range(iters)]

            def worker_b():
                start.wait()
                for k in range(iters):
                    go[k].wait()
                    if delay_us:
                        t_end = time.perf_counter() + delay_us * 1e-6
                        while time.perf_counter() < t_end:
                            pass
                    hb.run(params)
                    done_b[k].set()

            th = threading.Thread(target=worker_b)
            th.start()
            start.wait()
            for k in range(iters):
                go[k].set()
                ha.run(params)
                done_b[k].wait()
            th.join()
        return best(f)

    print("variant                         ms per step (1000 pairs, results of both halves in host memory)")
    print("one batch (product path)        %.4f" % best(one_batch))
    print("two halves back to back         %.4f" % back_to_back())
    print("two halves side by side         %.4f" % two_streams(0))
    for d in (40, 60, 80, 100):
        print("two halves, B %3d us after A    %.4f" % (d, two_streams(d)))
    print("one batch (again)               %.4f" % best(one_batch))
    for b in (whole, ha, hb):
        b.free()
    ca.close(); cb.close()


if __name__ == "__main__":
    main()
