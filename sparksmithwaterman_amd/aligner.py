"""Context / Batch: thin object wrappers over the C ABI (include/swmi.h)."""
import ctypes as C
import weakref

from . import _capi
from ._capi import Params, Timing, StreamStats, check

DEFAULT_SCORES = (5, -3, -4)              # Distribution.java:36  {match, mismatch, gap}
DEFAULT_TYPES = ("a", "i", "d", "-")      # Distribution.java:37


def make_params(align_scores=None, align_types=None, tie_mode=_capi.TIE_SERIAL):
    sc = DEFAULT_SCORES if align_scores is None else tuple(int(x) for x in align_scores)
    ty = DEFAULT_TYPES if align_types is None else tuple(align_types)
    if len(sc) != 3 or len(ty) != 4:
        raise ValueError("alignScores needs 3 entries and alignTypes 4")
    p = Params()
    p.match, p.mismatch, p.gap = sc
    p.tie_mode = tie_mode
    p.types = b"".join(_capi.as_bytes(t)[:1] for t in ty)
    return p


class Context:
    """One GPU + one HIP stream (swmi_ctx).  Use one per host thread."""

    def __init__(self, device=0):
        self._lib = _capi.load()
        h = C.c_void_p()
        check(self._lib.swmi_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def set_option(self, name, value):
        check(self._lib.swmi_set_option(self._h, name.encode(), int(value)))

    def upload(self, refs, reads):
        """Sequences -> HBM.  refs/reads: lists of str or bytes."""
        return Batch(self, refs, reads)

    def stream(self, reads, params=None, slots=0, chunk_bytes=0):
        """A Stream aligning `reads` against references that arrive in chunks (swmi_stream_*)."""
        return Stream(self, reads, params, slots, chunk_bytes)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.swmi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """refs x reads resident on the GPU (swmi_batch); pair index = ref * n_reads + read."""

    _owned = True

    @classmethod
    def _view(cls, lib, handle, n_refs, n_reads, owner):
        """results-only batch owned by a Stream.  The view keeps the Stream alive (swmi_stream_close frees every result
        batch), and Stream.close() sets its handle to None: the library answers a NULL batch with SWMI_ERR_INVALID, so a
        late call raises SwmiError instead of touching freed memory."""
        b = cls.__new__(cls)
        b._ctx, b._lib, b._h, b.n_refs, b.n_reads, b._owned = None, lib, handle, n_refs, n_reads, False
        b._owner = owner
        return b

    def __init__(self, ctx, refs, reads):
        self._ctx = ctx
        self._lib = ctx._lib
        self.n_refs, self.n_reads = len(refs), len(reads)
        rb, ro = _capi.pack(refs)
        qb, qo = _capi.pack(reads)
        h = C.c_void_p()
        check(self._lib.swmi_batch_upload(ctx._h, rb, ro, self.n_refs, qb, qo, self.n_reads, C.byref(h)))
        self._h = h

    def run(self, params=None):
        """Fill + traceback on the GPU for every pair, results to the host.  Synchronous."""
        p = params if params is not None else make_params()
        check(self._lib.swmi_batch_run(self._ctx._h, self._h, C.byref(p)))
        return self

    def run_async(self, params=None):
        """Starts the same run on the context's own host thread and returns at once; wait() completes it."""
        p = params if params is not None else make_params()
        self._async_params = p                     # (copied by the library before the call returns; kept for clarity)
        check(self._lib.swmi_batch_run_async(self._ctx._h, self._h, C.byref(p)))
        return self

    def wait(self):
        """Blocks until the run started by run_async() has finished; raises what it raised."""
        check(self._lib.swmi_batch_wait(self._ctx._h))
        return self

    def timing(self):
        t = Timing()
        check(self._lib.swmi_batch_timing(self._h, C.byref(t)))
        return t

    def pipeline_mode(self):
        v = C.c_int()
        check(self._lib.swmi_batch_mode(self._h, C.byref(v)))
        return v.value

    # ---- per pair -------------------------------------------------------------------
    def score(self, pair):
        v = C.c_int32()
        check(self._lib.swmi_pair_score(self._h, pair, C.byref(v)))
        return v.value

    def n_alignments(self, pair):
        n, f = C.c_uint64(), C.c_uint32()
        check(self._lib.swmi_pair_n_alignments(self._h, pair, C.byref(n), C.byref(f)))
        return n.value, f.value

    def alignment(self, pair, k, with_cell=False):
        b, ei, ej = C.c_int32(), C.c_int32(), C.c_int32()
        r, q, ln = C.c_char_p(), C.c_char_p(), C.c_uint32()
        check(self._lib.swmi_pair_alignment(self._h, pair, k, C.byref(b), C.byref(ei), C.byref(ej),
                                            C.byref(r), C.byref(q), C.byref(ln)))
        rec = (b.value, (r.value.decode("latin-1"), q.value.decode("latin-1")))
        return rec + ((ei.value, ej.value),) if with_cell else rec

    def alignments(self, pair, with_cell=False):
        n, _ = self.n_alignments(pair)
        return [self.alignment(pair, k, with_cell) for k in range(n)]

    def pair_results(self):
        """(scores int32[n_pairs], n_alignments uint64[n_pairs]) as numpy arrays, one native call."""
        import numpy as np
        n = self.n_refs * self.n_reads
        sc = np.empty(n, dtype=np.int32)
        na = np.empty(n, dtype=np.uint64)
        check(self._lib.swmi_batch_pair_results(self._h, sc.ctypes.data_as(C.POINTER(C.c_int32)),
                                                na.ctypes.data_as(C.POINTER(C.c_uint64)), n))
        return sc, na

    def scores(self):
        """every pair's score as a numpy int32 array (also after a scores_only run, which has no alignment counts)"""
        import numpy as np
        n = self.n_refs * self.n_reads
        sc = np.empty(n, dtype=np.int32)
        check(self._lib.swmi_batch_pair_results(self._h, sc.ctypes.data_as(C.POINTER(C.c_int32)), None, n))
        return sc

    def materialise_all(self):
        """Index the records and build every alignment string natively; returns (n_alignments, n_chars)."""
        na, nc = C.c_uint64(), C.c_uint64()
        check(self._lib.swmi_batch_materialise_all(self._h, C.byref(na), C.byref(nc)))
        return na.value, nc.value

    # ---- MapRef view -----------------------------------------------------------------
    def ref_total(self, ref):
        v = C.c_int32()
        check(self._lib.swmi_ref_total(self._h, ref, C.byref(v)))
        return v.value

    def ref_totals(self):
        """numpy int32 array of every reference's total (one native call)."""
        import numpy as np
        out = np.empty(self.n_refs, dtype=np.int32)
        check(self._lib.swmi_ref_totals(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), self.n_refs))
        return out

    def ref_match_sites(self, ref):
        n = C.c_uint64()
        check(self._lib.swmi_ref_n_match_sites(self._h, ref, C.byref(n)))
        out = []
        b, r, q, ln = C.c_int32(), C.c_char_p(), C.c_char_p(), C.c_uint32()
        for k in range(n.value):
            check(self._lib.swmi_ref_match_site(self._h, ref, k, C.byref(b), C.byref(r), C.byref(q), C.byref(ln)))
            out.append((b.value, (r.value.decode("latin-1"), q.value.decode("latin-1"))))
        return out

    def ref_sites_packed(self, ref_lo=0, ref_hi=None):
        """MapRef's output of references ref_lo .. ref_hi-1 in two native calls (sizes, then data): a list of
        (total, n_degenerate, [(begin, (refAligned, readAligned)), ...]) -- what swmi_ref_sites_packed hands a JNI binding."""
        import numpy as np
        ref_hi = self.n_refs if ref_hi is None else ref_hi
        n = ref_hi - ref_lo
        ns, nb = C.c_uint64(), C.c_uint64()
        check(self._lib.swmi_ref_sites_packed(self._h, ref_lo, ref_hi, None, None, None, None, None, None, 0, None, 0,
                                              C.byref(ns), C.byref(nb)))
        totals = np.empty(max(n, 1), dtype=np.int32)
        deg = np.empty(max(n, 1), dtype=np.uint64)
        first = np.empty(n + 1, dtype=np.uint64)
        begins = np.empty(max(ns.value, 1), dtype=np.int32)
        lens = np.empty(max(ns.value, 1), dtype=np.uint32)
        off = np.empty(max(ns.value, 1), dtype=np.uint64)
        blob = np.empty(max(nb.value, 1), dtype=np.uint8)
        P = C.POINTER
        check(self._lib.swmi_ref_sites_packed(
            self._h, ref_lo, ref_hi, totals.ctypes.data_as(P(C.c_int32)), deg.ctypes.data_as(P(C.c_uint64)),
            first.ctypes.data_as(P(C.c_uint64)), begins.ctypes.data_as(P(C.c_int32)), lens.ctypes.data_as(P(C.c_uint32)),
            off.ctypes.data_as(P(C.c_uint64)), ns.value, blob.ctypes.data_as(C.c_void_p), nb.value, C.byref(ns), C.byref(nb)))
        raw = blob.tobytes()
        out = []
        for r in range(n):
            sites = []
            for s in range(int(first[r]), int(first[r + 1])):
                o, ln = int(off[s]), int(lens[s])
                sites.append((int(begins[s]), (raw[o:o + ln].decode("latin-1"), raw[o + ln:o + 2 * ln].decode("latin-1"))))
            out.append((int(totals[r]), int(deg[r]), sites))
        return out

    def free(self):
        if getattr(self, "_h", None) and self._owned:
            self._lib.swmi_batch_free(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    """Reads resident, references streamed through the GPU in chunks (include/swmi.h: swmi_stream_*)."""

    def __init__(self, ctx, reads, params=None, slots=0, chunk_bytes=0):
        self._ctx, self._lib = ctx, ctx._lib
        self.n_reads = len(reads)
        qb, qo = _capi.pack(reads)
        p = params if params is not None else make_params()
        h = C.c_void_p()
        check(self._lib.swmi_stream_open(ctx._h, C.byref(p), qb, qo, self.n_reads, int(slots), int(chunk_bytes), C.byref(h)))
        self._h = h
        self._views = []                  # weak references to the result views handed out by chunks()

    def push(self, refs):
        rb, ro = _capi.pack(refs)
        check(self._lib.swmi_stream_push(self._h, rb, ro, len(refs)))
        return self

    def push_file(self, path, delimiter=">gi", parse_threads=0):
        import os
        check(self._lib.swmi_stream_push_file(self._h, os.fspath(path).encode(), delimiter.encode("latin-1"), int(parse_threads)))
        return self

    def finish(self):
        check(self._lib.swmi_stream_finish(self._h))
        return self

    def n_refs(self):
        return self._lib.swmi_stream_n_refs(self._h)

    def chunks(self):
        """[(first_ref, Batch view), ...] in reference order"""
        out = []
        for k in range(self._lib.swmi_stream_n_chunks(self._h)):
            b, first = C.c_void_p(), C.c_uint64()
            check(self._lib.swmi_stream_chunk(self._h, k, C.byref(b), C.byref(first)))
            n_pairs = self._lib.swmi_batch_n_pairs(b)
            v = Batch._view(self._lib, b, n_pairs // max(self.n_reads, 1), self.n_reads, self)
            self._views.append(weakref.ref(v))
            out.append((first.value, v))
        return out

    def totals(self):
        import numpy as np
        n = self.n_refs()
        out = np.empty(n, dtype=np.int32)
        check(self._lib.swmi_stream_totals(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return out

    def metadata(self, ref):
        buf = C.create_string_buffer(4096)
        check(self._lib.swmi_stream_metadata(self._h, ref, buf, 4096))
        return buf.value.decode("latin-1")

    def stats(self):
        st = StreamStats()
        check(self._lib.swmi_stream_get_stats(self._h, C.byref(st)))
        return st

    def close(self):
        if getattr(self, "_h", None):
            for w in getattr(self, "_views", []):      # the native close frees every result batch: no view may outlive it
                v = w()
                if v is not None:
                    v._h = None
            self._views = []
            self._lib.swmi_stream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
