"""The oracle against the known-answer vectors and against its independent Python twin.

The reference ships no tests or fixtures and cannot be run here
(no JVM): parity is "unpinned"; these KATs were derived by hand from the cited Java.
"""
import random

import pytest

from oracle import sw_oracle as orc
from oracle import sw_oracle_py as opy

REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"   # EngineerData.java:23
READ_80 = "AATTTTAGTCTCTCCCTACCCTTTTGGACAGAGCTTCCTGTCCTCTCATTTCACAGGTTATGCAACAGAGGGTTCTGTGT"  # EngineerData.java:26
READ_20 = "ACTGACTGACTGACTGACTG"   # EngineerData.java:29


def _norm(alns):
    return [[b, r, q] for (b, (r, q)) in alns]


def test_kats_c_oracle(kats):
    for k in kats:
        if k.get("H"):
            score, alns, H, T = orc.opt_alignments((k["ref"], k["read"]), k["scores"], k.get("types", "aid-").encode("latin-1"),
                                                   k["tie_mode"], matrices=True)
            assert H == k["H"], k["name"]
            if k.get("T"):
                # only cells with H > 0 are ever read by the traceback; compare those
                for i, row in enumerate(k["T"]):
                    for j, ch in enumerate(row):
                        if H[i][j] > 0:
                            assert T[i][j] == ch, (k["name"], i, j)
        else:
            score, alns = orc.opt_alignments((k["ref"], k["read"]), k["scores"], k.get("types", "aid-").encode("latin-1"), k["tie_mode"])
        assert score == k["score"], k["name"]
        assert _norm(alns) == k["alignments"], k["name"]


def test_kats_python_twin(kats):
    for k in kats:
        score, alns = opy.opt_alignments((k["ref"], k["read"]), tuple(k["scores"]), tuple(k.get("types", "aid-")),
                                         strict=bool(k["tie_mode"]))
        assert score == k["score"], k["name"]
        assert _norm(alns) == k["alignments"], k["name"]


def test_kat3_map_ref_sort(kats):
    k = [x for x in kats if x["name"] == "KAT-3-serial"][0]
    total, (ref, sites) = orc.map_ref((">gi|x", k["ref"]), [k["read"]], k["scores"])
    assert total == 2
    assert _norm(sites) == k["map_ref_sorted"]
    total2, (_, sites2) = opy.map_ref((">gi|x", k["ref"]), [k["read"]], tuple(k["scores"]))
    assert (total2, _norm(sites2)) == (total, _norm(sites))


@pytest.mark.parametrize("tie_mode", [0, 1])
def test_c_vs_python_random(tie_mode):
    rng = random.Random(1234 + tie_mode)
    for trial in range(150):
        alpha = rng.choice(["AC", "ACGT", "ACGTN", "acgtACGT"])
        n = rng.randint(0 if tie_mode == 0 else 1, 24)
        m = rng.randint(0 if tie_mode == 0 else 2, 12)
        ref = "".join(rng.choice(alpha) for _ in range(n))
        read = "".join(rng.choice(alpha) for _ in range(m))
        sc = rng.choice([(5, -3, -4), (1, -1, -1), (2, -1, -2), (1, 0, 0), (3, -2, 0)])
        a = orc.opt_alignments((ref, read), sc, b"aid-", tie_mode)
        b = opy.opt_alignments((ref, read), sc, ("a", "i", "d", "-"), strict=bool(tie_mode))
        assert a[0] == b[0], (ref, read, sc)
        assert _norm(a[1]) == _norm(b[1]), (ref, read, sc)


def test_case_insensitive_match_keeps_original_case():
    score, alns = orc.opt_alignments(("acgt", "ACGT"))
    assert score == 20
    assert alns == [(1, ("acgt", "ACGT"))]


def test_engineerdata_periodic_many_ties():
    # EngineerData.java:118 builds references as REF repeated: every period holds a tied maximum
    ref = REF * 5
    score, alns = orc.opt_alignments((ref, REF[10:50]))
    assert score == 200
    assert [a[0] for a in alns] == [11 + 80 * k for k in range(5)]
    s20, a20 = orc.opt_alignments((REF * 3, READ_20))
    p20 = opy.opt_alignments((REF * 3, READ_20))
    assert (s20, a20) == (p20[0], [(b, tuple(x)) for b, x in p20[1]])


def test_map_ref_multiple_reads_matches_python():
    ref = (">gi|ref0", REF * 2)
    reads = [READ_20, REF[5:35], READ_80[:30], ""]
    a = orc.map_ref(ref, reads)
    b = opy.map_ref(ref, reads)
    assert a[0] == b[0]
    assert _norm(a[1][1]) == _norm(b[1][1])


def test_bench_leg_counts_cells():
    r = orc.bench([REF * 2, REF], [READ_20, READ_80], nthreads=2)
    assert r["cells"] == (160 + 80) * (20 + 80)
    assert r["seconds"] > 0


def test_latin1_uppercase_rule():
    # Character.toUpperCase: e-acute (0xE9) -> E-acute (0xC9); the division sign 0xF7 and y-diaeresis 0xFF stay apart
    assert orc.opt_alignments(("\xe9", "\xc9"))[0] == 5
    assert orc.opt_alignments(("\xf7", "\xd7"))[0] == 0
    assert orc.opt_alignments(("\xff", "\xdf"))[0] == 0
    assert opy.opt_alignments(("\xe9", "\xc9"))[0] == 5
    assert opy.opt_alignments(("\xf7", "\xd7"))[0] == 0


def test_engineerdata_goldens_are_what_the_generator_writes():
    """tests/golden/engineerdata_small.json is data produced by tools/gen_engineerdata_golden.py (C oracle, required equal
    from the Python twin): the committed file must be exactly what the generator builds today."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_engineerdata_golden", os.path.join(root, "tools", "gen_engineerdata_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(root, "tests", "golden", "engineerdata_small.json")) as f:
        assert json.load(f) == gen.build()


def test_map_ref_goldens_both_oracles(map_ref_goldens):
    for g in map_ref_goldens:
        t, (_, sites) = orc.map_ref((">gi|ref0", g["ref"]), g["reads"], g["scores"], b"aid-", g["tie_mode"])
        assert (t, _norm(sites)) == (g["total"], g["match_sites"]), g["name"]
        t2, (_, s2) = opy.map_ref((">gi|ref0", g["ref"]), g["reads"], tuple(g["scores"]), ("a", "i", "d", "-"), strict=bool(g["tie_mode"]))
        assert (t2, _norm(s2)) == (g["total"], g["match_sites"]), g["name"]
