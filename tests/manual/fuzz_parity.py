#!/usr/bin/env python3
"""Randomised parity soak: random batches (lengths, alphabets, scores, tie modes, planted copies and repeats) through the
default pipeline and through sw_tfused_kernel (option tfused), every pair against the oracle.
    python tests/manual/fuzz_parity.py [seconds] [seed]"""
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
ctxs = {}
for name, tf in (("default", -1), ("tfused", 1)):
    c = sw.Context(0)
    c.set_option("tfused", tf)
    ctxs[name] = c
t_end = time.time() + budget
rounds = pairs = 0
while time.time() < t_end:
    alpha = rng.choice(["ACGT", "ACGT", "ACGTN", "AC", "acgtACGT"])
    rnd = lambda n: "".join(rng.choice(alpha) for _ in range(n))     # noqa: E731
    n_reads = rng.randint(1, 3)
    reads = [rnd(rng.choice([rng.randint(1, 40), rng.randint(60, 200), rng.randint(200, 256)])) for _ in range(n_reads)]
    refs = []
    for _ in range(rng.randint(2, 12)):
        n = rng.choice([rng.randint(1, 200), rng.randint(200, 1200), rng.randint(1200, 2560)])
        r = rnd(n)
        if rng.random() < 0.5:                       # plant copies of a read (tied maxima, possibly far apart)
            q = rng.choice(reads)
            for _ in range(rng.randint(1, 3)):
                at = rng.randint(0, max(0, len(r) - 1))
                r = r[:at] + q + r[at:]
            r = r[:2560]
        if rng.random() < 0.15:
            r = (rnd(rng.randint(3, 40)) * 200)[:rng.randint(50, 2000)]     # periodic: many ties
        refs.append(r)
    scores = rng.choice([(5, -3, -4), (1, -1, -1), (2, -1, -2), (7, -8, -1), (3, -2, -3), (5, -4, -8)])
    tie = rng.randint(0, 1)
    want = {}
    for name, ctx in ctxs.items():
        b = ctx.upload(refs, reads).run(sw.make_params(scores, ("a", "i", "d", "-"), tie))
        for r, ref in enumerate(refs):
            for q, read in enumerate(reads):
                key = (r, q)
                if key not in want:
                    want[key] = orc.opt_alignments((ref, read), scores, b"aid-", tie)
                es, ea = want[key]
                pair = r * len(reads) + q
                n, flags = b.n_alignments(pair)
                ok = b.score(pair) == es and n == len(ea) and (flags & sw.PAIR_DEGENERATE or n > 3000 or b.alignments(pair) == ea)
                if not ok:
                    print("MISMATCH", name, "seed", seed, "round", rounds, "pair", pair, "scores", scores, "tie", tie, len(ref), len(read), b.score(pair), es, n, len(ea), flush=True)
                    print(" ref", ref, "\n read", read, flush=True)
                    sys.exit(1)
        b.free()
    rounds += 1
    pairs += len(refs) * len(reads)
print("fuzz ok: %d rounds, %d pairs x 2 pipelines, seed %d" % (rounds, pairs, seed))
