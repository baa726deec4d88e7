"""Host-side mirror of the reference's I/O function objects.

  InOutOps.GetReads().call(file, delimiter)      -> [read, ...]                 src/sw/InOutOps.java:49,60-88
  InOutOps.GetRefSeqs().call(file, delimiter)    -> [[metadata, sequence], ...] src/sw/InOutOps.java:100,115-168
  InOutOps.GetOutputStr().call(reads, (nums, max, execTime), opt) -> str        src/sw/InOutOps.java:226,244-288
  InOutOps.PrintStrToFile().call(filepath, data) -> bool                        src/sw/InOutOps.java:182,196-218
  DirectoryCrawler(root).hasNext()/next()                                       src/sw/DirectoryCrawler.java

Parsing is done by the native mmap reader in libswmi.so (include/swmi_io.h); `read_reads_packed` /
`read_refs_packed` hand back the packed blob + offsets that Batch uploads without building Python strings.
"""
import ctypes as C
import os

from . import _capi

NEWLINE = os.linesep      # InOutOps.NEWLINE = System.lineSeparator()   :38
TAB = "\t"                # InOutOps.TAB                                :39
DELIMITER = ">gi"         # Distribution.java:46

_P = C.c_void_p
IO_SYMBOLS = [
    ("swmi_io_read_reads", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_P)]),
    ("swmi_io_read_refs", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_P)]),
    ("swmi_seqset_count", C.c_uint32, [_P]),
    ("swmi_seqset_bytes", C.POINTER(C.c_uint8), [_P]),
    ("swmi_seqset_offsets", C.POINTER(C.c_uint64), [_P]),
    ("swmi_seqset_metadata", C.c_char_p, [_P, C.c_uint32]),
    ("swmi_seqset_free", None, [_P]),
]
_bound = False


def _lib():
    global _bound
    lib = _capi.load()
    if not _bound:
        for name, res, args in IO_SYMBOLS:
            f = getattr(lib, name)
            f.restype = res
            f.argtypes = args
        _bound = True
    return lib


class SeqSet:
    """Packed result of the native reader: .blob (bytes), .offsets (list), .metadata (list of str)."""

    def __init__(self, handle, with_meta):
        lib = _lib()
        try:
            n = lib.swmi_seqset_count(handle)
            offs = lib.swmi_seqset_offsets(handle)
            self.offsets = [offs[k] for k in range(n + 1)]
            total = self.offsets[-1]
            self.blob = C.string_at(lib.swmi_seqset_bytes(handle), total) if total else b""
            self.metadata = ([lib.swmi_seqset_metadata(handle, k).decode("latin-1") for k in range(n)]
                             if with_meta else [""] * n)
        finally:
            lib.swmi_seqset_free(handle)

    def __len__(self):
        return len(self.offsets) - 1

    def sequences(self):
        return [self.blob[self.offsets[k]:self.offsets[k + 1]].decode("latin-1") for k in range(len(self))]


def _path(f):
    return os.fspath(f).encode()


def read_reads_packed(file, delimiter=DELIMITER):
    h = _P()
    _capi.check(_lib().swmi_io_read_reads(_path(file), delimiter.encode("latin-1"), C.byref(h)))
    return SeqSet(h, False)


def read_refs_packed(file, delimiter=DELIMITER):
    h = _P()
    _capi.check(_lib().swmi_io_read_refs(_path(file), delimiter.encode("latin-1"), C.byref(h)))
    return SeqSet(h, True)


class InOutOps:
    NEWLINE = NEWLINE
    TAB = TAB

    class IsMetadata:
        def call(self, line, delimiter):       # InOutOps.java:405-411
            return len(line) >= len(delimiter) and line[:len(delimiter)] == delimiter

    class GetReads:
        def call(self, file, delimiter=DELIMITER):
            return read_reads_packed(file, delimiter).sequences()

    class GetRefSeqs:
        def call(self, file, delimiter=DELIMITER):
            s = read_refs_packed(file, delimiter)
            return [[m, q] for m, q in zip(s.metadata, s.sequences())]

    class GetOutputStr:
        def call(self, reads, data, opt):
            """data = ((numRefs, numReads), maxScore, execTimeMs); opt = [((metadata, seq), sites), ...]"""
            nums, max_score, exec_time = data
            out = []
            out.append("Execution Time = %s ms%s" % (exec_time, NEWLINE))          # :249
            out.append(NEWLINE)
            out.append("# Reference Sequences = %s%s" % (nums[0], NEWLINE))         # :253
            out.append("# Reads = %s%s" % (nums[1], NEWLINE))
            out.append(NEWLINE)
            out.append("Input:" + NEWLINE)                                          # :258
            for read in reads:
                out.append(read + NEWLINE)
            out.append(NEWLINE)
            out.append("Maximum alignment score = %s" % max_score)                  # :264
            out.append(NEWLINE)
            for seq, sites in opt:                                                  # :268-285
                out.append("Reference:" + NEWLINE)
                out.append(seq[0] + NEWLINE)
                out.append(seq[1] + NEWLINE)
                out.append(NEWLINE)
                for begin, aligned in sites:
                    out.append(TAB + "Index = %s%s" % (begin, NEWLINE))
                    out.append(TAB + aligned[0] + NEWLINE)
                    out.append(TAB + aligned[1] + NEWLINE)
                    out.append(NEWLINE)
            return "".join(out)

    class PrintStrToFile:
        def call(self, filepath, data):        # the directory must exist, the file is overwritten (:200-208)
            try:
                with open(filepath, "w", newline="", encoding="latin-1") as f:
                    f.write(data)
                return True
            except OSError as e:
                print("IOException on writing to file", e)
                return False


class DirectoryCrawler:
    """Depth-first iterator over the FILES below `root` (DirectoryCrawler.java:25-139).

    `File.listFiles()` order is OS dependent in the reference; here children are visited in sorted name order
    so that runs are reproducible."""

    def __init__(self, root):
        if not os.path.exists(root):
            raise FileNotFoundError("Root directory not found")     # the reference prints this and exits (:30-34)
        self._flat = list(self._walk(root))
        self._i = -1

    def _walk(self, d):
        for name in sorted(os.listdir(d)):
            p = os.path.join(d, name)
            if os.path.isdir(p):
                for x in self._walk(p):
                    yield x
            else:
                yield p

    def hasNext(self):
        self._i += 1
        return self._i < len(self._flat)

    def next(self):
        return self._flat[self._i]
