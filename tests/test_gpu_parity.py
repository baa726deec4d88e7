"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Scores are Java ints and alignments are strings: equality is exact, no tolerance anywhere.
The oracle itself is "parity unpinned" (the reference ships no fixtures and there is no JVM to run it);
see oracle/sw_oracle.c.
"""
import random

import pytest

import sparksmithwaterman_amd as sw
from sparksmithwaterman_amd import synth
from oracle import sw_oracle as orc

pytestmark = pytest.mark.gpu

REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"   # EngineerData.java:23
READ_80 = "AATTTTAGTCTCTCCCTACCCTTTTGGACAGAGCTTCCTGTCCTCTCATTTCACAGGTTATGCAACAGAGGGTTCTGTGT"  # EngineerData.java:26
READ_20 = "ACTGACTGACTGACTGACTG"   # EngineerData.java:29


@pytest.fixture(scope="module", params=[(1, 1, 0, 0, 0), (2, 1, 0, 0, 0), (0, 1, 0, 0, 0), (1, 0, 0, 0, 0), (-1, 1, -1, -1, -1), (1, 1, 1, 0, 0),
                                        (1, 0, 1, 0, 0), (1, 1, 0, 1, 0), (1, 0, -1, 1, 0), (1, 1, 0, 0, 1), (1, 0, -1, -1, 1),
                                        (-1, 1, -1, -1, -1, 0)],
                ids=["mode1-winmax", "mode2-events", "mode0-field", "mode1-d2h-copy", "automatic", "mode1-split-traceback",
                     "mode1-split-d2h-copy", "mode1-resident", "mode1-resident-d2h-copy", "mode1-tfused", "mode1-tfused-d2h-copy",
                     "automatic-host-strings"])
def ctx(request):
    """Every kernel pipeline (include/swmi.h, swmi_set_option "mode"), results written straight to pinned host
    memory or fetched by a copy, the mode-1 traceback with one workgroup per pair or split per window / alignment
    (option "tb_split"), small pairs handled whole by one wavefront with the direction field in LDS (option "resident":
    0 never, 1 wherever it fits), the usual pair swept in the transposed layout and traced back by the same wavefront
    (option "tfused": 0 never, 1 every pair that qualifies); -1 lets the library choose."""
    c = sw.Context(0)
    c.set_option("mode", request.param[0])
    c.set_option("zero_copy", request.param[1])
    c.set_option("tb_split", request.param[2])
    c.set_option("resident", request.param[3])
    c.set_option("tfused", request.param[4])
    # the two aligned strings are written by the traceback kernels (the default); 0: the host builds them from the 2-bit ops
    c.set_option("device_strings", request.param[5] if len(request.param) > 5 else 1)
    c.test_resident = request.param[3]
    c.test_tfused = request.param[4]
    yield c
    c.close()


def _types(t):
    return b"".join(x.encode() for x in t) if not isinstance(t, bytes) else t


def check_batch(ctx, refs, reads, scores=(5, -3, -4), types=("a", "i", "d", "-"), tie=0, cells=True):
    b = ctx.upload(refs, reads).run(sw.make_params(scores, types, tie))
    try:
        for r, ref in enumerate(refs):
            for q, read in enumerate(reads):
                pair = r * len(reads) + q
                es, ea = orc.opt_alignments((ref, read), scores, _types(types), tie, with_cells=cells and tie == 0)
                assert b.score(pair) == es, (r, q)
                n, flags = b.n_alignments(pair)
                assert n == len(ea), (r, q, n, len(ea))
                if flags & sw.PAIR_DEGENERATE:
                    assert es == 0 and all(a[:2] == (0, ("", "")) for a in ea)
                    if n:
                        assert b.alignment(pair, 0) == (0, ("", ""))
                        assert b.alignment(pair, n - 1, True) == (0, ("", ""), (len(read), len(ref)))
                    continue
                got = b.alignments(pair, with_cell=cells and tie == 0)
                assert got == ea, (r, q, ref, read, scores, tie)
            t, (_, sites) = orc.map_ref((">gi|r%d" % r, ref), reads, scores, _types(types), tie)
            assert b.ref_total(r) == t
            if not any(b.n_alignments(r * len(reads) + q)[1] & sw.PAIR_DEGENERATE and
                       b.n_alignments(r * len(reads) + q)[0] > 5000 for q in range(len(reads))):
                assert b.ref_match_sites(r) == sites
    finally:
        b.free()


def test_kats(ctx, kats):
    for k in kats:
        b = ctx.upload([k["ref"]], [k["read"]]).run(sw.make_params(k["scores"], tuple(k.get("types", "aid-")), k["tie_mode"]))
        assert b.score(0) == k["score"], k["name"]
        assert [[x[0], x[1][0], x[1][1]] for x in b.alignments(0)] == k["alignments"], k["name"]
        if "map_ref_sorted" in k:
            assert [[x[0], x[1][0], x[1][1]] for x in b.ref_match_sites(0)] == k["map_ref_sorted"]
        b.free()


def test_map_ref_goldens(ctx, map_ref_goldens):
    """EngineerData-shaped MapRef goldens (tests/golden/engineerdata_small.json): total and the stably sorted match sites"""
    for g in map_ref_goldens:
        b = ctx.upload([g["ref"]], g["reads"]).run(sw.make_params(g["scores"], tuple("aid-"), g["tie_mode"]))
        assert b.ref_total(0) == g["total"], g["name"]
        assert [[x[0], x[1][0], x[1][1]] for x in b.ref_match_sites(0)] == g["match_sites"], g["name"]
        b.free()


def test_mirror_classes_match_reference_call_shapes(ctx):
    score, alns = sw.SmithWaterman.OptAlignments(ctx).call(["ACGT", "CG"], [5, -3, -4], ["a", "i", "d", "-"])
    assert (score, alns) == (10, [(2, ("CG", "CG"))])
    score, alns = sw.DistributedSW.OptAlignments(ctx).call(["AAAA", "AACA"], [1, -1, -1], ["a", "i", "d", "-"])
    assert score == 2 and alns[3] == (2, ("AA_A", "AACA"))
    ref = (">gi|ref0", REF * 2)
    reads = [READ_20, REF[5:35], READ_80[:30], ""]
    algo = ([5, -3, -4], ["a", "i", "d", "-"])
    total, (ref_out, sites) = sw.Distribution.MapRef(ctx).call((ref, reads, algo))
    et, (_, es) = orc.map_ref(ref, reads)
    assert ref_out is ref and total == et and sites == es
    tuples = sw.Distribution.CombineReadsToRef().call([ref, (">gi|ref1", REF[::-1])], reads, algo)
    res = sw.Distribution.MapPartition(ctx).call(tuples)
    assert [r[0] for r in res] == [orc.map_ref(t[0], reads)[0] for t in tuples]
    red = sw.Distribution.ReduceMax()
    for t, v in res:
        red.add(t, v)
    assert red.result()[0] == max(r[0] for r in res)


@pytest.mark.parametrize("tie", [0, 1])
def test_random_small_alphabets_many_ties(ctx, tie):
    rng = random.Random(99 + tie)
    for trial in range(12):
        alpha = rng.choice(["AC", "ACGT", "ACGTN", "acgtACGT", "AX-"])
        refs = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 70))) for _ in range(6)]
        reads = ["".join(rng.choice(alpha) for _ in range(rng.randint(2 if tie else 1, 40))) for _ in range(5)]
        sc = rng.choice([(5, -3, -4), (1, -1, -1), (2, -1, -2), (1, 0, 0), (3, -2, 0), (2, -3, -1)])
        check_batch(ctx, refs, reads, sc, tie=tie)


def test_read_lengths_across_row_classes(ctx):
    # rows per lane R = 1,2,3,4 and the multi-strip path (m > 256), ragged reference lengths
    rng = random.Random(5)
    reads = ["".join(rng.choice("ACGT") for _ in range(m)) for m in (1, 63, 64, 65, 128, 129, 150, 192, 193, 256, 257, 300, 520)]
    refs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (1, 15, 16, 17, 100, 333)]
    refs.append(reads[6][10:120] + "ACGTTGCA" * 10)
    check_batch(ctx, refs, reads)


def test_empty_and_degenerate(ctx):
    check_batch(ctx, ["CCCC", "", "ACGT", "GGGGGGGGGG"], ["AA", "", "CG"])
    check_batch(ctx, ["CCCCCCCCCCCCCCCCCCCCCCCCCCCCCC" * 4], ["A" * 70])


def test_non_acgt_and_case(ctx):
    check_batch(ctx, ["acgtNNNNacgtRYKM", "ACGTNNNNACGTRYKM", "nnnn\xe9\xc9"], ["ACGTNNNN", "acgtnnnnacgt", "N\xe9\xc9n"])


def test_engineerdata_periodic_overflowing_cell_list(ctx):
    # EngineerData.java:118 repeats REF: every period is a tied maximum. 100 periods > cell_cap (64)
    check_batch(ctx, [REF * 100, REF * 3], [REF[10:50], READ_20, READ_80])
    ctx.set_option("cell_cap", 4)
    try:
        check_batch(ctx, [REF * 9], [REF[10:50], READ_20])
    finally:
        ctx.set_option("cell_cap", 64)


def test_arena_growth(ctx):
    ctx.set_option("arena_words_per_pair", 1)
    try:
        rng = random.Random(17)
        refs = ["".join(rng.choice("ACGT") for _ in range(400)) for _ in range(40)]
        check_batch(ctx, refs, [refs[3][50:200]])
    finally:
        ctx.set_option("arena_words_per_pair", 48)


def test_unusual_scores_generic_path(ctx):
    rng = random.Random(3)
    refs = ["".join(rng.choice("ACGT") for _ in range(90)) for _ in range(4)]
    reads = ["".join(rng.choice("ACGT") for _ in range(33)) for _ in range(3)]
    check_batch(ctx, refs, reads, (300, -200, -150))       # scores outside int8: compare path, not the profile
    check_batch(ctx, refs, reads, (2, 1, 1), cells=True)   # positive mismatch / gap


def test_duplicate_align_types_rejected(ctx):
    with pytest.raises(sw.SwmiError) as e:
        ctx.upload(["ACGT"], ["CG"]).run(sw.make_params((5, -3, -4), ("a", "a", "d", "-")))
    assert e.value.code == -5


def test_config1_headline_batch_scores_and_winner(ctx):
    refs, reads = synth.config_1k()
    b = ctx.upload(refs, reads).run()
    rng = random.Random(1)
    sample = [0] + rng.sample(range(1, len(refs)), 24)
    for r in sample:
        es, ea = orc.opt_alignments((refs[r], reads[0]))
        assert b.score(r) == es
        assert b.alignments(r) == ea
    totals = [b.ref_total(r) for r in range(len(refs))]
    assert max(range(len(refs)), key=lambda r: totals[r]) == 0      # the read was cut from reference 0
    # checksum of all scores against the oracle's multi-threaded pass over the same batch
    ob = orc.bench(refs, reads, nthreads=8)
    assert sum(totals) == ob["sum_score"]
    assert sum(b.n_alignments(r)[0] for r in range(len(refs))) == ob["sum_aln"]
    b.free()


def test_chunked_workspace_streaming(ctx):
    # a workspace cap far below the batch's need forces several launch chunks (how the 10^6-reference config streams)
    refs, reads = synth.config_ncbi(300, read_len=150, seed=2)
    ctx.set_option("max_workspace_bytes", 1 << 20)
    try:
        b = ctx.upload(refs, reads).run()
        assert b.timing().fill_launches > 3
        rng = random.Random(7)
        for r in rng.sample(range(len(refs)), 40):
            es, ea = orc.opt_alignments((refs[r], reads[0]))
            assert b.score(r) == es and b.alignments(r) == ea
        ob = orc.bench(refs, reads, nthreads=8)
        assert int(b.ref_totals().astype("int64").sum()) == ob["sum_score"]
        assert sum(b.n_alignments(r)[0] for r in range(len(refs))) == ob["sum_aln"]
        b.free()
    finally:
        ctx.set_option("max_workspace_bytes", 32 << 30)


def test_long_pairs_multi_strip(ctx):
    # configs[4] shape at reduced length: 3000 x 3000 pairs need 12 strips of 256 read rows each
    refs, reads = synth.config_long(n_pairs=2, length=3000, seed=4)
    b = ctx.upload(refs, reads).run()
    for r in range(2):
        for q in range(2):
            es, ea = orc.opt_alignments((refs[r], reads[q]))
            assert b.score(r * 2 + q) == es
            assert b.alignments(r * 2 + q) == ea
    b.free()


def test_ncbi_shaped_batch_checksums(ctx):
    # configs[2] shape (log-normal reference lengths, median 1,609 bp) at 4,000 references: checksum of all scores and
    # alignment counts against the oracle's multi-threaded pass, plus the winner
    refs, reads = synth.config_ncbi(4000, read_len=150, seed=2)
    b = ctx.upload(refs, reads).run()
    totals = b.ref_totals()
    ob = orc.bench(refs, reads, nthreads=32)
    assert int(totals.astype("int64").sum()) == ob["sum_score"]
    assert sum(b.n_alignments(r)[0] for r in range(len(refs))) == ob["sum_aln"]
    w = int(totals.argmax())
    es, ea = orc.opt_alignments((refs[w], reads[0]))
    assert b.score(w) == es and b.alignments(w) == ea
    b.free()


def _planted(rng, read, n, starts, alphabet="T"):
    """A reference of `alphabet` filler with exact copies of `read` at the given 0-based starts (read has no filler base)."""
    ref = [rng.choice(alphabet) for _ in range(n)]
    for s in starts:
        ref[s:s + len(read)] = list(read)
    return "".join(ref)


def test_tied_maxima_across_checkpoint_windows(ctx):
    """Exact copies of the read planted in an all-T reference: every copy ends in a tied maximum cell, so the number of
    alignments, the windows they end in and their distance from the reference's ends are under control.  Exercises the
    mode-1 traceback's teams (1, 2, 3-4 and more than 4 alignments per pair), candidates at the very first / last columns,
    copies that straddle 32-step checkpoint windows, several maxima inside one window, and both workgroup shapes
    (8 waves per pair for batches of <= 512 pairs, 4 otherwise)."""
    rng = random.Random(20261004)
    read100 = "".join(rng.choice("ACG") for _ in range(100))
    read10 = "".join(rng.choice("ACG") for _ in range(10))
    read200 = "".join(rng.choice("ACG") for _ in range(200))       # R = 4 rows per lane
    cases = [
        (read100, 700, [0]),                          # the alignment starts in column 1
        (read100, 700, [600]),                        # ... and ends in the last column
        (read100, 700, [13, 300]),                    # two candidates
        (read100, 700, [5, 210, 420]),                # three: one walker each, no helpers in a 4-wave workgroup
        (read100, 900, [0, 130, 260, 390, 520, 650, 780]),   # more alignments than walkers
        (read100, 333, [27, 156]),                    # ends at steps 126/255: last column of a window
        (read10, 300, [3, 14, 25, 36, 200]),          # several maxima inside one window (more cells than candidates)
        (read10, 40, [0, 30]),                        # reference shorter than two windows
        (read200, 1000, [100, 500, 777]),
        (read200, 260, [60]),
    ]
    refs, reads_for = [], []
    for read, n, starts in cases:
        refs.append(_planted(rng, read, n, starts))
        reads_for.append(read)
    # one batch per read (a batch is refs x reads); small batches run 8 waves per pair
    for read in (read100, read10, read200):
        rs = [r for r, q in zip(refs, reads_for) if q is read]
        for tie in (0, 1):
            check_batch(ctx, rs, [read], tie=tie)
    # the same references repeated past 512 pairs: 4 waves per pair
    rs = [r for r, q in zip(refs, reads_for) if q is read100]
    many = (rs * 110)[:560]
    b = ctx.upload(many, [read100]).run(sw.make_params())
    try:
        want = {}
        for k, ref in enumerate(many):
            if ref not in want:
                want[ref] = orc.opt_alignments((ref, read100), (5, -3, -4), b"aid-", 0, with_cells=True)
            es, ea = want[ref]
            assert b.score(k) == es
            assert b.alignments(k, with_cell=True) == ea, k
    finally:
        b.free()


def test_long_reads_mixed_with_short_ones(ctx):
    """Reads of one, two and three strips (256 rows each) in ONE batch, one of them outside ACGT: in mode 1 the long
    ones are swept one wavefront per strip (pipelined through the seam rows) next to the ordinary one-wave pairs."""
    rng = random.Random(77)
    refs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (1500, 900, 1201)]

    def cut(ref, m, alphabet="ACGT", every=11):
        s = list(ref[100:100 + m])
        for k in range(0, m, every):
            s[k] = rng.choice(alphabet)
        return "".join(s)

    reads = [cut(refs[0], 700), cut(refs[1], 150), cut(refs[2], 300, "ACGTN", 7), cut(refs[0], 257, every=5)]
    for tie in (0, 1):
        check_batch(ctx, refs, reads, tie=tie)


def test_fast_symbols_and_int4_score_bounds(ctx):
    """The fast cell stream covers the symbols A,C,G,T,N,U,R,Y (any case) with match/mismatch inside int4; one step outside
    either (another letter, match = 8, mismatch = -9) must take the compare-and-select variant and agree as well."""
    rng = random.Random(99)

    def seqs(alphabet, n, lens):
        return ["".join(rng.choice(alphabet) for _ in range(rng.choice(lens))) for _ in range(n)]

    refs = seqs("ACGTN", 6, (90, 300, 700)) + seqs("acgtnURY", 3, (200, 333))
    reads = seqs("ACGTN", 2, (40, 150)) + seqs("ACGTUY", 1, (300,))
    for scores in ((5, -3, -4), (7, -8, -2), (1, -1, -1), (8, -3, -4), (5, -9, -4)):
        check_batch(ctx, refs, reads, scores=scores)
    check_batch(ctx, refs + seqs("ACGTNK", 2, (120,)), reads, tie=1)      # K is not a fast symbol


def test_run_async_matches_run(ctx):
    """swmi_batch_run_async + swmi_batch_wait: the same results as the blocking run, one run in flight per context."""
    refs, reads = synth.config_1k(n_refs=24, ref_len=400, read_len=150)
    b = ctx.upload(refs, reads)
    b.run()
    want = [(b.score(k), b.alignments(k)) for k in range(len(refs))]
    for _ in range(3):
        b.run_async()
        with pytest.raises(sw.SwmiError):
            b.run_async()                      # a second run while one is in flight is refused ...
        b.wait()                               # ... and does not disturb the first
        assert [(b.score(k), b.alignments(k)) for k in range(len(refs))] == want
    with pytest.raises(sw.SwmiError):
        b.wait()                               # nothing in flight
    b.free()


def test_column_chunks_of_long_references(ctx):
    """Few pairs, long references: the mode-1 sweep of a pair is cut into column chunks, one wavefront each, every chunk
    re-deriving its left context from a halo no positive-score path can span (swmi_device.h: ColItem).  Checked against
    the oracle in full -- scores, every tied maximum, every alignment -- for forced chunk counts and the automatic one,
    random and periodic references (EngineerData.java:118: REF repeated, one tied maximum per period), reads of one strip and
    of two and three (EngineerData.java:87-104: read lengths up to 500)."""
    rng = random.Random(42)
    rnd = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (9000, 20011, 4097)]
    reads = [rnd[0][4000:4150], rnd[1][10:90], READ_80, rnd[1][19000:19250]]
    # reads of several strips (> 256 rows): every column chunk is a strip pipeline of its own (swmi_device.h: StripItem) --
    # 520 rows cut out of a reference, 700 rows (three strips) with substitutions and a gap, EngineerData's periodic read
    long3 = rnd[1][5000:5300] + "A" + rnd[1][5300:5340] + rnd[1][5350:5700]
    long3 = "".join("ACGT"[("ACGT".index(c) + 1) % 4] if i % 41 == 0 else c for i, c in enumerate(long3))
    reads += [rnd[0][2000:2520], long3, (READ_80 * 7)[:300]]
    refs = rnd + [REF * 130, "T" * 5000 + reads[0] + "T" * 3000 + reads[0][:120] + "T" * 900]
    for chunks in (0, 2, 7, 64, 1):
        ctx.set_option("col_chunks", chunks)
        try:
            b = ctx.upload(refs, reads).run()
            if chunks > 1 and b.pipeline_mode() == 1:
                assert b.timing().col_chunks >= 2 * len(reads)
            if chunks == 1:
                assert b.timing().col_chunks == 0
            b.free()
            check_batch(ctx, refs, reads)
            check_batch(ctx, refs[:2], reads[:2], scores=(2, -1, -3), tie=1)
        finally:
            ctx.set_option("col_chunks", 0)


def test_config3_shape_many_reads_totals_winners_topk(ctx):
    """configs[3] shape on one GPU: many reads x NCBI-shaped references -> per-reference totals (MapRef, Distribution.java:
    403-436) -> the driver's max-with-ties reduce (:600-613) and the top-K the multi-GPU path exchanges.  Every pair's
    score and alignment count against the oracle, full match-site lists on a sample of references."""
    import numpy as np
    from sparksmithwaterman_amd import distributed as swd
    refs, reads = synth.config_multi_read(500, 64, read_len=150, seed=3)
    b = ctx.upload(refs, reads).run()
    ob = orc.bench(refs, reads, nthreads=16, per_pair=True)
    sc, na = b.pair_results()
    assert [int(x) for x in sc] == ob["pair_score"]
    assert [int(x) for x in na] == ob["pair_naln"]
    totals = b.ref_totals()
    want_tot = np.asarray(ob["pair_score"], dtype=np.int64).reshape(len(refs), len(reads)).sum(axis=1)
    assert [int(x) for x in totals] == [int(x) for x in want_tot]
    # the driver's reduce: running maximum with ties, over MapRef results
    red = sw.Distribution.ReduceMax()
    for r in range(len(refs)):
        red.add(int(totals[r]), ([">gi|ref%d" % r, refs[r]], None))
    mx, opt = red.result()
    assert mx == int(want_tot.max())
    assert sorted(int(v[0][0][7:]) for v in opt) == [int(r) for r in np.flatnonzero(want_tot == want_tot.max())]
    # what the sharded path computes: two shards reduced as two ranks would, and the top-K merge
    lo0, hi0 = swd.shard_bounds(len(refs), 0, 2)
    assert swd.global_max_with_ties(totals[lo0:hi0], range(lo0, hi0))[0] <= mx
    topk = swd.global_top_k(totals, range(len(refs)), 8)
    order = sorted(range(len(refs)), key=lambda r: (-int(want_tot[r]), r))[:8]
    assert topk == [(int(want_tot[r]), r) for r in order]
    # full MapRef output (match sites stably sorted by begin) on the winners and a sample
    rng = random.Random(9)
    for r in set(order[:3] + rng.sample(range(len(refs)), 5)):
        t, (_, sites) = orc.map_ref((">gi|r", refs[r]), reads)
        assert int(totals[r]) == t
        assert b.ref_match_sites(r) == sites
    b.free()


def test_resident_pairs_engineerdata_shapes(ctx):
    """The reference's own benchmark shapes (EngineerData.java:51-224): 80 bp reads against 400 bp periodic references, one
    tied maximum per period.  In mode 1 such pairs are handled by sw_resident_pairs_kernel (two sweeps inside LDS, every
    alignment walked by its own lane); checked in full against the oracle, with a batch large enough to fill the chip, more
    tied maxima than lanes in one pair, and reads that do not match at all."""
    rng = random.Random(31)
    refs = [REF * 5, REF * 5, (REF * 5)[3:393], REF[::-1] * 4, "".join(rng.choice("ACGT") for _ in range(400)), "G" * 300]
    reads = [READ_80, REF[7:87], READ_20, REF[40:80] + REF[:40], "ACGT" * 10, "T" * 30]
    check_batch(ctx, refs, reads)
    check_batch(ctx, refs[:3], reads[:3], scores=(1, -1, -1), tie=1)
    b = ctx.upload(refs, reads).run()
    if b.pipeline_mode() == 1 and ctx.test_resident == 1:
        assert b.timing().resident_pairs > 0
    b.free()
    many = [REF * 5] * 3000 + [REF[::-1] * 5] * 100
    b = ctx.upload(many, [READ_80, READ_20]).run()
    if b.pipeline_mode() == 1 and ctx.test_resident != 0:
        assert b.timing().resident_pairs >= 3100           # (automatic: launches of at least 256 pairs)
    sc, na = b.pair_results()
    want = {}
    for k in (0, 1, 2999, 3000, 3099):
        for q, read in enumerate((READ_80, READ_20)):
            es, ea = want.setdefault((many[k], read), orc.opt_alignments((many[k], read)))
            assert int(sc[k * 2 + q]) == es and b.alignments(k * 2 + q) == ea
    assert len(set(int(x) for x in sc[0:6000:2])) == 1 and len(set(int(x) for x in na[0:6000:2])) == 1
    b.free()


def test_transposed_fused_blocks_ties_and_long_paths(ctx):
    """What sw_tfused_kernel (option "tfused") has that the small cases do not reach: tied maxima in DISTANT blocks of one
    reference (block tasks), more tied maxima in one block than its cell list holds (passes), several alignments per block
    (walk items read another wavefront's tile), paths longer than a block (the walk re-sweeps the block on the left), both tie
    modes, references at the edges of the columns-per-lane classes.  Every score and alignment against the oracle; the other
    context variants run the same batch through their own kernels."""
    rng = random.Random(77)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))     # noqa: E731
    read = rnd(150)
    planted = rnd(300) + read + rnd(700) + read + rnd(600) + read[:149] + rnd(40)          # the read three times, far apart
    near = rnd(500) + read + "ACGT" + read + rnd(300)                                      # ... and twice within one block
    refs = [planted, near, (REF * 30)[:2000], "ACGTTGCA" * 250, rnd(2047), rnd(2049), rnd(2560), rnd(127), rnd(129), read, read[::-1] * 8]
    reads = [read, REF[10:60], rnd(150), "ACGTTGCA" * 12]
    check_batch(ctx, refs, reads)
    check_batch(ctx, refs[:5], reads[:2] + reads[3:], tie=1)
    # long reads and a cheap gap: paths of several hundred columns leave their 320-column block
    long_reads = [rnd(256), (planted[100:330] + "TTGACCA")[:237]]
    check_batch(ctx, [planted, rnd(1500)], long_reads, scores=(5, -3, -1))
    check_batch(ctx, [planted], long_reads[:1], scores=(7, -8, -1), tie=1)
    b = ctx.upload(refs, reads).run()
    if b.pipeline_mode() == 1 and ctx.test_tfused == 1:
        assert b.timing().tfused_pairs >= len(refs) * len(reads) - 4            # (all but the pairs with an empty... none here: every pair qualifies)
    b.free()


def test_config4_full_size_pair(ctx):
    """configs[4] at full size: one 10 kbp x 10 kbp pair (40 strips of 256 read rows, 10^8 cells) against the oracle --
    score, every tied maximum, every alignment string (full size had only been checked run to run)."""
    refs, reads = synth.config_long(n_pairs=1, length=10000, seed=4)
    b = ctx.upload(refs, reads).run()
    es, ea = orc.opt_alignments((refs[0], reads[0]))
    assert b.score(0) == es
    assert b.alignments(0) == ea
    b.free()


def test_many_tied_maxima_in_every_pair_of_a_large_launch(ctx):
    """A launch large enough for one workgroup per pair (>= 64 pairs) whose pairs each carry more tied maxima than a wave
    has lanes: the records come back unranked (SWMI_RANK_BY_CELL) and the host orders them by cell, in both tie orders."""
    ref = REF * 70                                           # 5600 bp, one tied maximum per period
    refs = [ref, ref[40:] + ref[:40], ref[::-1]] * 24        # 72 pairs per read
    for tie in (0, 1):
        b = ctx.upload(refs, [REF[10:50]]).run(sw.make_params((5, -3, -4), ("a", "i", "d", "-"), tie))
        for k in (0, 1, 2, 71):
            es, ea = orc.opt_alignments((refs[k], REF[10:50]), (5, -3, -4), b"aid-", tie)
            assert b.score(k) == es and b.n_alignments(k)[0] == len(ea)
            assert b.alignments(k) == ea
        b.free()
