#!/usr/bin/env python3
"""Randomised soak of two batches in flight on one GPU (one context each, started alternately with swmi_batch_run_async, as
bench.py's step loop does): a fresh random batch per slot every few steps, every retired step against the oracle -- scores,
alignment counts, and every alignment string of a sample of pairs.
    python tests/manual/fuzz_in_flight.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sparksmithwaterman_amd as sw           # noqa: E402
from oracle import sw_oracle as orc           # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
ctxs = [sw.Context(0), sw.Context(0)]


def new_job():
    alpha = rng.choice(["ACGT", "ACGT", "ACGTN", "AC"])
    rnd = lambda n: "".join(rng.choice(alpha) for _ in range(n))     # noqa: E731
    reads = [rnd(rng.choice([rng.randint(20, 100), rng.randint(100, 256), rng.randint(257, 400)])) for _ in range(rng.randint(1, 2))]
    refs = []
    for _ in range(rng.choice([rng.randint(1, 12), rng.randint(50, 300), rng.randint(300, 1200)])):
        r = rnd(rng.randint(100, 2200))
        if rng.random() < 0.6:
            q = rng.choice(reads)
            at = rng.randint(0, len(r) - 1)
            r = r[:at] + q[:rng.randint(len(q) // 2, len(q))] + r[at:]
        refs.append(r)
    ob = orc.bench(refs, reads, nthreads=8, per_pair=True)
    return refs, reads, ob["pair_score"], ob["pair_naln"]


jobs = [None, None]
bs = [None, None]
in_flight = [False, False]
steps = pairs = 0


def retire(i):
    global pairs
    bs[i].wait()
    refs, reads, want_s, want_n = jobs[i]
    sc, na = bs[i].pair_results()
    if [int(x) for x in sc] != want_s or [int(x) for x in na] != want_n:
        print("MISMATCH (scores / counts) seed", seed, "step", steps, "slot", i, flush=True)
        sys.exit(1)
    for _ in range(3):
        p = rng.randrange(len(refs) * len(reads))
        if want_n[p] <= 200 and bs[i].alignments(p) != orc.opt_alignments((refs[p // len(reads)], reads[p % len(reads)]))[1]:
            print("MISMATCH (alignments) seed", seed, "step", steps, "slot", i, "pair", p, flush=True)
            sys.exit(1)
    pairs += len(want_s)
    in_flight[i] = False


t_end = time.time() + budget
while time.time() < t_end:
    i = steps % 2
    if in_flight[i]:
        retire(i)
    if jobs[i] is None or rng.random() < 0.3:
        if bs[i] is not None:
            bs[i].free()
        jobs[i] = new_job()
        bs[i] = ctxs[i].upload(jobs[i][0], jobs[i][1])
    bs[i].run_async()
    in_flight[i] = True
    steps += 1
    if steps % 200 == 0:
        print("steps %d, pairs checked %d" % (steps, pairs), flush=True)
for i in range(2):
    if in_flight[i]:
        retire(i)
print("OK: %d steps with two batches in flight, %d pairs checked, seed %d" % (steps, pairs, seed))
