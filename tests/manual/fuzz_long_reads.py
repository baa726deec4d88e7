#!/usr/bin/env python3
"""Randomised parity soak of the strip pipeline and its column chunks (reads of 257 .. 1100 rows, few pairs, references of
0.6 .. 14 kbp, forced and automatic chunk counts, both tie modes, reads cut out of the references and mutated, periodic
references): every pair against the oracle -- score, alignment count, every alignment.
    python tests/manual/fuzz_long_reads.py [seconds] [seed]"""
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
ctx = sw.Context(0)
t_end = time.time() + budget
rounds = pairs = chunked = 0
while time.time() < t_end:
    alpha = rng.choice(["ACGT", "ACGT", "ACGTN", "AC", "acgtACGT", "ACGTK"])
    rnd = lambda n: "".join(rng.choice(alpha) for _ in range(n))     # noqa: E731
    refs = []
    for _ in range(rng.randint(1, 5)):
        n = rng.choice([rng.randint(600, 3000), rng.randint(3000, 8000), rng.randint(8000, 14000)])
        refs.append((rnd(rng.randint(5, 90)) * 400)[:n] if rng.random() < 0.15 else rnd(n))
    reads = []
    for _ in range(rng.randint(1, 3)):
        m = rng.choice([rng.randint(257, 300), rng.randint(300, 520), rng.randint(520, 1100)])
        src = rng.choice(refs)
        if rng.random() < 0.8 and len(src) > m + 10:
            at = rng.randint(0, len(src) - m - 1)
            q = list(src[at:at + m])
            for _ in range(rng.randint(0, m // 12)):                 # substitutions, insertions, deletions
                p = rng.randint(0, len(q) - 1)
                c = rng.random()
                if c < 0.6:
                    q[p] = rng.choice(alpha)
                elif c < 0.8:
                    q.insert(p, rng.choice(alpha))
                elif len(q) > 258:
                    del q[p]
            reads.append("".join(q)[:1100])
        else:
            reads.append(rnd(m))
    scores = rng.choice([(5, -3, -4), (5, -3, -4), (1, -1, -1), (2, -1, -2), (7, -8, -1), (3, 0, -3), (5, -4, -8), (9, -3, -4)])
    tie = rng.randint(0, 1)
    chunks = rng.choice([0, 0, 0, 2, 5, 17, 64, 1])
    ctx.set_option("col_chunks", chunks)
    # the other pipelines share the strip code (fill_pair / fill_block16): the direction field (0), event-tracked maxima (2),
    # the fused and the split traceback, results by copy instead of zero-copy
    ctx.set_option("mode", rng.choice([-1, -1, 1, 1, 0, 2]))
    ctx.set_option("tb_split", rng.choice([-1, -1, 0, 1]))
    ctx.set_option("zero_copy", rng.choice([1, 1, 0]))
    b = ctx.upload(refs, reads).run(sw.make_params(scores, ("a", "i", "d", "-"), tie))
    chunked += 1 if b.timing().col_chunks else 0
    for r, ref in enumerate(refs):
        for q, read in enumerate(reads):
            es, ea = orc.opt_alignments((ref, read), scores, b"aid-", tie)
            pair = r * len(reads) + q
            n, flags = b.n_alignments(pair)
            ok = b.score(pair) == es and n == len(ea) and (flags & sw.PAIR_DEGENERATE or n > 3000 or b.alignments(pair) == ea)
            if not ok:
                print("MISMATCH seed", seed, "round", rounds, "pair", pair, "scores", scores, "tie", tie, "col_chunks", chunks,
                      len(ref), len(read), b.score(pair), es, n, len(ea), flush=True)
                sys.exit(1)
            pairs += 1
    b.free()
    rounds += 1
    if rounds % 20 == 0:
        print("rounds %d, pairs %d, launches with column chunks %d" % (rounds, pairs, chunked), flush=True)
print("OK: %d rounds, %d pairs, %d launches with column chunks, seed %d" % (rounds, pairs, chunked, seed))
ctx.close()
