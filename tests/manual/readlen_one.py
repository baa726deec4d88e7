#!/usr/bin/env python3
"""One point of the read-length sweep, a few runs (for rocprofv3): python tests/manual/readlen_one.py [L]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparksmithwaterman_amd as sw      # noqa: E402
REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"
READ_80 = "AATTTTAGTCTCTCCCTACCCTTTTGGACAGAGCTTCCTGTCCTCTCATTTCACAGGTTATGCAACAGAGGGTTCTGTGT"
L = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = sw.Context(0)
b = ctx.upload([REF * 50], [(READ_80 * 7)[:L]] * 5)
for _ in range(6):
    b.run()
b.free()
ctx.close()
