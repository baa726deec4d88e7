"""ctypes loader for oracle/sw_oracle.c -- TEST INFRASTRUCTURE ONLY ("parity unpinned").

Builds oracle/_build/libsw_oracle.so with gcc on first use if it is missing.
Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsw_oracle.so")
_lib = None

TIE_SERIAL = 0   # SmithWaterman.java semantics
TIE_STRICT = 1   # DistributedSW.java semantics

DEFAULT_SCORES = (5, -3, -4)
DEFAULT_TYPES = b"aid-"


def build(force=False):
    src = os.path.join(_HERE, "sw_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.sw_oracle_align.restype = C.c_void_p
        L.sw_oracle_align.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64,
                                      C.POINTER(C.c_int32), C.c_char_p, C.c_int, C.c_int]
        L.sw_oracle_map_ref.restype = C.c_void_p
        L.sw_oracle_map_ref.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.POINTER(C.c_int64),
                                        C.c_int64, C.POINTER(C.c_int32), C.c_char_p, C.c_int]
        L.sw_oracle_free.argtypes = [C.c_void_p]
        L.sw_oracle_score.restype = C.c_int32
        L.sw_oracle_score.argtypes = [C.c_void_p]
        L.sw_oracle_n_aln.restype = C.c_int64
        L.sw_oracle_n_aln.argtypes = [C.c_void_p]
        for nm in ("begin", "end_i", "end_j"):
            f = getattr(L, "sw_oracle_aln_" + nm)
            f.restype = C.c_int32
            f.argtypes = [C.c_void_p, C.c_int64]
        for nm in ("ref", "read"):
            f = getattr(L, "sw_oracle_aln_" + nm)
            f.restype = C.c_char_p
            f.argtypes = [C.c_void_p, C.c_int64]
        L.sw_oracle_H.restype = C.POINTER(C.c_int32)
        L.sw_oracle_H.argtypes = [C.c_void_p]
        L.sw_oracle_T.restype = C.POINTER(C.c_char)
        L.sw_oracle_T.argtypes = [C.c_void_p]
        L.sw_oracle_bench.restype = C.c_double
        L.sw_oracle_bench.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int64,
                                      C.c_char_p, C.POINTER(C.c_int64), C.c_int64,
                                      C.POINTER(C.c_int32), C.c_char_p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1")


def _collect(L, h, with_cells):
    n = L.sw_oracle_n_aln(h)
    out = []
    for k in range(n):
        rec = (L.sw_oracle_aln_begin(h, k),
               (L.sw_oracle_aln_ref(h, k).decode("latin-1"), L.sw_oracle_aln_read(h, k).decode("latin-1")))
        if with_cells:
            rec = rec + ((L.sw_oracle_aln_end_i(h, k), L.sw_oracle_aln_end_j(h, k)),)
        out.append(rec)
    return out


def opt_alignments(seqs, align_scores=DEFAULT_SCORES, align_types=DEFAULT_TYPES,
                   tie_mode=TIE_SERIAL, with_cells=False, matrices=False):
    """SmithWaterman.OptAlignments.call(seqs={ref,read}, alignScores, alignTypes).

    Returns (score, [(begin, (refAligned, readAligned)), ...]); with matrices=True
    also returns (H, T) as nested lists for eyeball diffs (InOutOps.PrintMatrices).
    """
    L = lib()
    ref, read = _b(seqs[0]), _b(seqs[1])
    sc = (C.c_int32 * 3)(*align_scores)
    h = L.sw_oracle_align(ref, len(ref), read, len(read), sc, _b(align_types), tie_mode, int(matrices))
    if not h:
        raise MemoryError("sw_oracle_align failed")
    try:
        score = L.sw_oracle_score(h)
        alns = _collect(L, h, with_cells)
        if matrices:
            m, n = len(read), len(ref)
            Hp, Tp = L.sw_oracle_H(h), L.sw_oracle_T(h)
            H = [[Hp[i * (n + 1) + j] for j in range(n + 1)] for i in range(m + 1)]
            T = [[Tp[i * (n + 1) + j].decode("latin-1") for j in range(n + 1)] for i in range(m + 1)]
            return score, alns, H, T
        return score, alns
    finally:
        L.sw_oracle_free(h)


def pack(seqs):
    """list of str/bytes -> (concatenated bytes, int64 offsets array of len+1)."""
    bs = [_b(s) for s in seqs]
    off = (C.c_int64 * (len(bs) + 1))()
    t = 0
    for k, b in enumerate(bs):
        off[k] = t
        t += len(b)
    off[len(bs)] = t
    return b"".join(bs), off


def map_ref(ref, reads, align_scores=DEFAULT_SCORES, align_types=DEFAULT_TYPES, tie_mode=TIE_SERIAL):
    """Distribution.MapRef.call: ref=(metadata, sequence) -> (total, (ref, matchSites))."""
    L = lib()
    rb = _b(ref[1])
    blob, off = pack(reads)
    sc = (C.c_int32 * 3)(*align_scores)
    h = L.sw_oracle_map_ref(rb, len(rb), blob, off, len(reads), sc, _b(align_types), tie_mode)
    if not h:
        raise MemoryError("sw_oracle_map_ref failed")
    try:
        return L.sw_oracle_score(h), (ref, _collect(L, h, False))
    finally:
        L.sw_oracle_free(h)


def bench(refs, reads, align_scores=DEFAULT_SCORES, align_types=DEFAULT_TYPES,
          tie_mode=TIE_SERIAL, nthreads=1, reps=1, per_pair=False):
    """Times the full CPU path over refs x reads, `reps` passes by a thread pool started before the clock.
    Returns dict(seconds, cells (all passes), sum_score, sum_aln (one pass)); per_pair=True adds the lists
    pair_score / pair_naln (pair = ref * n_reads + read)."""
    L = lib()
    rblob, roff = pack(refs)
    qblob, qoff = pack(reads)
    sc = (C.c_int32 * 3)(*align_scores)
    ss, sa, cc = C.c_int64(), C.c_int64(), C.c_int64()
    npair = len(refs) * len(reads)
    ps = (C.c_int32 * max(npair, 1))() if per_pair else None
    pa = (C.c_int64 * max(npair, 1))() if per_pair else None
    sec = L.sw_oracle_bench(rblob, roff, len(refs), qblob, qoff, len(reads), sc, _b(align_types),
                            tie_mode, nthreads, reps, C.byref(ss), C.byref(sa), C.byref(cc), ps, pa)
    out = {"seconds": sec, "cells": cc.value, "sum_score": ss.value // reps, "sum_aln": sa.value // reps}
    if per_pair:
        out["pair_score"] = list(ps[:npair])
        out["pair_naln"] = list(pa[:npair])
    return out
