#!/usr/bin/env python3
"""small, chatty run of sw_tfused_kernel (debugging aid): python tests/manual/tfused_small.py [n_refs]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sparksmithwaterman_amd as sw
from sparksmithwaterman_amd import synth
from oracle import sw_oracle as orc
n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
refs, reads = synth.config_1k(n_refs=n_refs, ref_len=2000, read_len=150)
print("inputs ready", flush=True)
ctx = sw.Context(0)
ctx.set_option("tfused", 1)
print("running", flush=True)
b = ctx.upload(refs, reads).run(sw.make_params((5, -3, -4), ("a", "i", "d", "-"), 0))
print("ran; tfused pairs", b.timing().tfused_pairs, flush=True)
bad = 0
for r, ref in enumerate(refs):
    es, ea = orc.opt_alignments((ref, reads[0]), (5, -3, -4), b"aid-", 0, with_cells=True)
    got = (b.score(r), b.alignments(r, with_cell=True))
    if got != (es, ea):
        bad += 1
        print("MISMATCH", r, got[0], es, len(got[1]), len(ea), flush=True)
print("mismatches", bad, flush=True)
