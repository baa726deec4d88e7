#!/usr/bin/env python3
"""sw_tfused_kernel against the oracle on the headline shape: every pair's score, alignment count and alignments.
    python tests/manual/tfused_check.py [n_refs] [ref_len] [read_len] [tie]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sparksmithwaterman_amd as sw           # noqa: E402
from sparksmithwaterman_amd import synth      # noqa: E402
from oracle import sw_oracle as orc           # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
read_len = int(sys.argv[3]) if len(sys.argv) > 3 else 150
tie = int(sys.argv[4]) if len(sys.argv) > 4 else 0
refs, reads = synth.config_1k(n_refs=n_refs, ref_len=ref_len, read_len=read_len)
ctx = sw.Context(0)
ctx.set_option("tfused", 1)
ctx.set_option("profiling", 1)
p = sw.make_params((5, -3, -4), ("a", "i", "d", "-"), tie)
b = ctx.upload(refs, reads).run(p)
t = b.timing()
print("tfused pairs %d of %d, rerun %d, fill %.4f ms, traceback %.4f ms" % (t.tfused_pairs, len(refs) * len(reads), t.rerun_pairs, t.fill_ms, t.traceback_ms))
t0 = time.time()
for _ in range(20):
    b.run(p)
print("ms per run: %.4f" % ((time.time() - t0) / 20 * 1e3), "fill %.4f tb %.4f" % (b.timing().fill_ms, b.timing().traceback_ms))
bad = 0
for r, ref in enumerate(refs):
    for q, read in enumerate(reads):
        pair = r * len(reads) + q
        es, ea = orc.opt_alignments((ref, read), (5, -3, -4), b"aid-", tie, with_cells=tie == 0)
        if b.score(pair) != es or b.alignments(pair, with_cell=tie == 0) != ea:
            bad += 1
            if bad < 5:
                print("MISMATCH pair", pair, b.score(pair), es, b.n_alignments(pair), len(ea))
                print("  got", b.alignments(pair, with_cell=tie == 0)[:2])
                print("  exp", ea[:2])
print("pairs compared %d, mismatches %d" % (len(refs) * len(reads), bad))
sys.exit(1 if bad else 0)
