"""CPU oracle for the Smith-Waterman hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See the header of oracle/sw_oracle.c ("parity unpinned").
"""
