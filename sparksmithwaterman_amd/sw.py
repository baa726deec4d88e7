"""Host-side mirror of the reference's operator interface for the hot path.

Same class names, argument meaning and return shapes as the Java function objects, so a
test written against the reference reads the same here:

  SmithWaterman.OptAlignments().call(seqs, alignScores, alignTypes)
        -> (score, [(begin, (refAligned, readAligned)), ...])      src/sw/SmithWaterman.java:35,62-92
  DistributedSW.OptAlignments().call(...)   same, '>' tie order       src/sw/DistributedSW.java:50,77-104
  Distribution.MapRef().call((ref, reads, (alignScores, alignTypes)))
        -> (total, (ref, matchSites))                               src/sw/Distribution.java:383,403-436
  Distribution.MapPartition().call(iterable of such tuples)        one native call per partition
  Distribution.CombineReadsToRef().call(refs, reads, algoArgs)      src/sw/Distribution.java:702-725
  Distribution.ReduceMax                                           control-path max-with-ties, :600-613, :647-666

All alignment work happens in libswmi.so on the GPU; nothing here computes a score.
"""
from . import _capi
from .aligner import Context, make_params, DEFAULT_SCORES, DEFAULT_TYPES

_default_ctx = {}


def default_context(device=0):
    c = _default_ctx.get(device)
    if c is None:
        c = _default_ctx[device] = Context(device)
    return c


def _algo(align_scores, align_types):
    return (DEFAULT_SCORES if align_scores is None else align_scores,
            DEFAULT_TYPES if align_types is None else align_types)


class SmithWaterman:
    class OptAlignments:
        """Function3<String[], int[], char[], Tuple2<Integer, ArrayList<Tuple2<Integer,String[]>>>>."""
        tie_mode = _capi.TIE_SERIAL

        def __init__(self, context=None):
            self._ctx = context

        def call(self, seqs, alignScores=None, alignTypes=None):
            ctx = self._ctx or default_context()
            sc, ty = _algo(alignScores, alignTypes)
            b = ctx.upload([seqs[0]], [seqs[1]])
            try:
                b.run(make_params(sc, ty, self.tie_mode))
                return b.score(0), b.alignments(0)
            finally:
                b.free()


class DistributedSW:
    class OptAlignments(SmithWaterman.OptAlignments):
        """Same recurrence with DistributedSW's strict '>' tie order and diagonal max-cell order."""
        tie_mode = _capi.TIE_STRICT


class Distribution:
    ALIGN_SCORES = DEFAULT_SCORES
    ALIGN_TYPES = DEFAULT_TYPES

    class CombineReadsToRef:
        def call(self, references, reads, algoArgs):
            return [(ref, reads, algoArgs) for ref in references]

    class MapPartition:
        """mapPartitions variant: every (ref, reads, algoArgs) tuple of a partition in ONE native call.

        Tuples that share the reads list and algoArgs (what CombineReadsToRef builds) are batched
        together; results come back in input order, each exactly what MapRef.call returns.
        """

        def __init__(self, context=None, tie_mode=_capi.TIE_SERIAL):
            self._ctx = context
            self._tie = tie_mode

        def call(self, tuples):
            tuples = list(tuples)
            out = [None] * len(tuples)
            groups = {}
            for idx, (ref, reads, algo) in enumerate(tuples):
                key = (id(reads), None if algo is None else (tuple(algo[0]), tuple(algo[1])))
                groups.setdefault(key, []).append(idx)
            ctx = self._ctx or default_context()
            for idxs in groups.values():
                _, reads, algo = tuples[idxs[0]]
                sc, ty = _algo(*(algo if algo is not None else (None, None)))
                b = ctx.upload([tuples[i][0][1] for i in idxs], list(reads))
                try:
                    b.run(make_params(sc, ty, self._tie))
                    for r, i in enumerate(idxs):
                        out[i] = (b.ref_total(r), (tuples[i][0], b.ref_match_sites(r)))
                finally:
                    b.free()
            return out

    class MapRef:
        """PairFunction<Tuple3<String[], ArrayList<String>, Tuple2<int[],char[]>>, Integer, Tuple2<...>>."""

        def __init__(self, context=None, tie_mode=_capi.TIE_SERIAL):
            self._mp = Distribution.MapPartition(context, tie_mode)

        def call(self, tuple3):
            return self._mp.call([tuple3])[0]

    class ReduceMax:
        """Running maximum with ties over MapRef results, then OptSeqsComp order.

        Control-path semantics (NoDistribution, Distribution.java:600-613; sort :621, :647-666).
        DistributeReference's own reduce takes `first()` of an RDD whose sorted copy was discarded
        (Distribution.java:341-342) and so reports the first reference's total, not the maximum;
        the intended (control) behaviour is what is implemented here.
        """

        def __init__(self):
            self.max = 0
            self.opt = []

        def add(self, total, value):
            if total > self.max:
                self.max = total
                self.opt = [value]
            elif total == self.max:
                self.opt.append(value)

        def result(self):
            return self.max, sorted(self.opt, key=lambda v: v[0][0])


# ------------------------------------------------------------------------------------------------
# file-level drivers: same positional ioArgs / algoArgs as the reference
# ------------------------------------------------------------------------------------------------
import time as _time

from . import io as _io


class _FileDriver:
    """Shared body of the two drivers below.  ioArgs = [refDir, inDir, delimiter, outDir, outName, outExt]
    (nulls keep the defaults, Distribution.java:267-281); algoArgs = (alignScores, alignTypes) or None (:284-292)."""

    OUT_FILE, OUT_EXT = "result", ".txt"                              # Distribution.java:40-41
    REF_DIR, IN_DIR = "/home/ubuntu/project/reference", "/home/ubuntu/project/input"   # :43-44
    OUT_DIR = "/home/ubuntu/project/output/reference"                 # :50

    def __init__(self, context=None, tie_mode=_capi.TIE_SERIAL):
        self._ctx = context
        self._tie = tie_mode

    def call(self, ioArgs=None, algoArgs=None):
        ref_dir, in_dir, delim = self.REF_DIR, self.IN_DIR, _io.DELIMITER
        out_dir, out_name, out_ext = self.OUT_DIR, self.OUT_FILE, self.OUT_EXT
        if ioArgs is not None and len(ioArgs) == 6:
            ref_dir = ioArgs[0] if ioArgs[0] is not None else ref_dir
            in_dir = ioArgs[1] if ioArgs[1] is not None else in_dir
            delim = ioArgs[2] if ioArgs[2] is not None else delim
            out_dir = ioArgs[3] if ioArgs[3] is not None else out_dir
            out_name = ioArgs[4] if ioArgs[4] is not None else out_name
            out_ext = ioArgs[5] if ioArgs[5] is not None else out_ext
        sc, ty = _algo(*(algoArgs if algoArgs is not None else (None, None)))
        ctx = self._ctx or default_context()
        params = make_params(sc, ty, self._tie)

        in_crawl = _io.DirectoryCrawler(in_dir)
        input_num = 0
        while in_crawl.hasNext():
            input_num += 1
            reads = _io.InOutOps.GetReads().call(in_crawl.next(), delim)
            num_refs = 0
            t0 = _time.time()
            red = Distribution.ReduceMax()                       # `int max = 0` + ties, Distribution.java:573,600-613
            ref_crawl = _io.DirectoryCrawler(ref_dir)
            while ref_crawl.hasNext():
                rs = _io.read_refs_packed(ref_crawl.next(), delim)
                num_refs += len(rs)
                seqs = rs.sequences()
                b = ctx.upload(seqs, reads).run(params)          # every reference of the file x every read: one batch
                try:
                    for r in range(len(rs)):
                        total = b.ref_total(r)
                        if total >= red.max:                     # match sites are only materialised for candidates
                            red.add(total, ([rs.metadata[r], seqs[r]], b.ref_match_sites(r)))
                finally:
                    b.free()
            exec_ms = int((_time.time() - t0) * 1000)
            mx, opt = red.result()                               # Collections.sort(opt, new OptSeqsComp())  :621
            text = _io.InOutOps.GetOutputStr().call(reads, ((num_refs, len(reads)), mx, exec_ms), opt)
            _io.InOutOps.PrintStrToFile().call("%s/%s%d%s" % (out_dir, out_name, input_num, out_ext), text)
        return None


class _NoDistribution(_FileDriver):
    """Distribution.NoDistribution.call (Distribution.java:482-634), alignments on the GPU."""
    OUT_DIR = "/home/ubuntu/project/output/control"               # :49


class _DistributeReference(_FileDriver):
    """Distribution.DistributeReference.call (Distribution.java:227-373) with the control path's reduce.

    The reference's own reduce (`sortByKey` result discarded, then `first()`, :341-342) reports the FIRST
    reference's total instead of the maximum; the intended semantics (running max with ties, :600-613) are used."""


Distribution.NoDistribution = _NoDistribution
Distribution.DistributeReference = _DistributeReference
