"""ctypes binding of libswmi.so (include/swmi.h).  No torch types cross this boundary.

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C sparksmithwaterman_amd/csrc``.
Loading fails loudly when it is missing: there is no Python or CPU implementation behind it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SWMI_LIB: another build of the same sources, e.g. the -DSWMI_CK_BLOCKS=4u variant of tools/build_variants.sh)
LIB_PATH = os.environ.get("SWMI_LIB") or os.path.join(_HERE, "lib", "libswmi.so")

TIE_SERIAL = 0
TIE_STRICT = 1
PAIR_DEGENERATE = 0x1


class SwmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("swmi error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32),
                ("tie_mode", C.c_int32), ("types", C.c_char * 4)]


class Timing(C.Structure):
    _fields_ = [("fill_ms", C.c_float), ("traceback_ms", C.c_float), ("d2h_ms", C.c_float),
                ("total_ms", C.c_float), ("fill_launches", C.c_uint32), ("rerun_pairs", C.c_uint32),
                ("cells", C.c_uint64), ("dir_bytes", C.c_uint64),
                ("strip_fallbacks", C.c_uint32), ("col_chunks", C.c_uint32),
                ("resident_pairs", C.c_uint32), ("tfused_pairs", C.c_uint32)]


class StreamStats(C.Structure):
    _fields_ = [("push_ms", C.c_double), ("parse_ms", C.c_double), ("upload_ms", C.c_double), ("run_ms", C.c_double),
                ("gpu_sweep_ms", C.c_double), ("gpu_traceback_ms", C.c_double), ("bytes", C.c_uint64), ("cells", C.c_uint64),
                ("chunks", C.c_uint32), ("pad", C.c_uint32)]


# every symbol include/swmi.h declares: (name, restype, argtypes)
_P = C.c_void_p
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
SYMBOLS = [
    ("swmi_abi_version", C.c_int, []),
    ("swmi_last_error", C.c_char_p, []),
    ("swmi_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("swmi_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("swmi_destroy", None, [_P]),
    ("swmi_default_params", None, [C.POINTER(Params)]),
    ("swmi_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("swmi_batch_upload", C.c_int, [_P, C.c_char_p, _u64p, C.c_uint32, C.c_char_p, _u64p, C.c_uint32, C.POINTER(_P)]),
    ("swmi_batch_run", C.c_int, [_P, _P, C.POINTER(Params)]),
    ("swmi_batch_run_async", C.c_int, [_P, _P, C.POINTER(Params)]),
    ("swmi_batch_wait", C.c_int, [_P]),
    ("swmi_batch_free", None, [_P, _P]),
    ("swmi_batch_timing", C.c_int, [_P, C.POINTER(Timing)]),
    ("swmi_batch_mode", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("swmi_batch_n_pairs", C.c_uint64, [_P]),
    ("swmi_pair_score", C.c_int, [_P, C.c_uint64, C.POINTER(C.c_int32)]),
    ("swmi_pair_n_alignments", C.c_int, [_P, C.c_uint64, _u64p, C.POINTER(C.c_uint32)]),
    ("swmi_pair_alignment", C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_uint32)]),
    ("swmi_batch_pair_results", C.c_int, [_P, C.POINTER(C.c_int32), _u64p, C.c_uint64]),
    ("swmi_batch_materialise_all", C.c_int, [_P, _u64p, _u64p]),
    ("swmi_ref_total", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_int32)]),
    ("swmi_ref_totals", C.c_int, [_P, C.POINTER(C.c_int32), C.c_uint32]),
    ("swmi_ref_n_match_sites", C.c_int, [_P, C.c_uint32, _u64p]),
    ("swmi_ref_match_site", C.c_int, [_P, C.c_uint32, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_uint32)]),
    ("swmi_ref_sites_packed", C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), _u64p, _u64p, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_uint32), _u64p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p, _u64p]),
    ("swmi_stream_open", C.c_int, [_P, C.POINTER(Params), C.c_char_p, _u64p, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(_P)]),
    ("swmi_stream_push", C.c_int, [_P, C.c_char_p, _u64p, C.c_uint32]),
    ("swmi_stream_push_file", C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_uint32]),
    ("swmi_stream_finish", C.c_int, [_P]),
    ("swmi_stream_n_refs", C.c_uint64, [_P]),
    ("swmi_stream_n_chunks", C.c_uint32, [_P]),
    ("swmi_stream_chunk", C.c_int, [_P, C.c_uint32, C.POINTER(_P), _u64p]),
    ("swmi_stream_totals", C.c_int, [_P, C.POINTER(C.c_int32), C.c_uint64]),
    ("swmi_stream_metadata", C.c_int, [_P, C.c_uint64, C.c_char_p, C.c_size_t]),
    ("swmi_stream_get_stats", C.c_int, [_P, C.POINTER(StreamStats)]),
    ("swmi_stream_close", None, [_P]),
    ("swmi_align_batch", C.c_int, [_P, C.POINTER(Params), C.c_char_p, _u64p, C.c_uint32, C.c_char_p, _u64p,
                                   C.c_uint32, C.POINTER(_P)]),
]

_lib = None


def load():
    """Load libswmi.so and bind every declared symbol.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(lib, name)      # AttributeError if a declared symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise SwmiError(rc, load().swmi_last_error().decode("utf-8", "replace"))


def as_bytes(s):
    return bytes(s) if isinstance(s, (bytes, bytearray, memoryview)) else s.encode("latin-1")


def pack(seqs):
    """list of str/bytes -> (blob, uint64 offsets[len+1])."""
    bs = [as_bytes(s) for s in seqs]
    off = (C.c_uint64 * (len(bs) + 1))()
    t = 0
    for k, b in enumerate(bs):
        off[k] = t
        t += len(b)
    off[len(bs)] = t
    return b"".join(bs), off
