#!/usr/bin/env python3
"""Writes tests/golden/engineerdata_small.json: EngineerData-shaped golden vectors.

The reference's own benchmark material is three sequence literals -- REF, READ_80, READ_20
(src/metrics/EngineerData.java:23,26,29) -- and recipes that repeat REF (:51-224): periodic references with a tied
maximum per period.  The cases here are REF x 5 against READ_80 and READ_20, in both tie modes, plus the MapRef view of
both reads together.  Expected outputs come from the C oracle (oracle/sw_oracle.c) and this script REFUSES to write them
unless the independently written Python twin (oracle/sw_oracle_py.py) returns exactly the same: the file pins the two
restatements to each other, so a later edit of either cannot drift silently ("parity unpinned" still holds: nothing the
reference ships pins the oracles themselves).

    python tools/gen_engineerdata_golden.py          # rewrites the file; tests/test_oracle.py checks it is up to date
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"   # EngineerData.java:23
READ_80 = "AATTTTAGTCTCTCCCTACCCTTTTGGACAGAGCTTCCTGTCCTCTCATTTCACAGGTTATGCAACAGAGGGTTCTGTGT"  # EngineerData.java:26
READ_20 = "ACTGACTGACTGACTGACTG"   # EngineerData.java:29
SCORES = [5, -3, -4]               # Distribution.java:36


def build():
    from oracle import sw_oracle as orc
    from oracle import sw_oracle_py as opy
    ref = REF * 5
    cases = []
    for rname, read in (("READ_80", READ_80), ("READ_20", READ_20)):
        for tie in (0, 1):
            sc, al = orc.opt_alignments((ref, read), SCORES, b"aid-", tie)
            sp, ap = opy.opt_alignments((ref, read), tuple(SCORES), ("a", "i", "d", "-"), strict=bool(tie))
            if (sc, al) != (sp, ap):
                raise SystemExit("the C oracle and the Python twin disagree on REF x 5 / %s / tie %d" % (rname, tie))
            cases.append({"name": "REFx5-%s-%s" % (rname, "strict" if tie else "serial"), "ref": ref, "read": read, "scores": SCORES,
                          "tie_mode": tie, "score": sc, "alignments": [[b, r, q] for (b, (r, q)) in al]})
    maps = []
    for tie in (0, 1):
        t, (_, sites) = orc.map_ref((">gi|ref0", ref), [READ_80, READ_20], SCORES, b"aid-", tie)
        tp, (_, sp) = opy.map_ref((">gi|ref0", ref), [READ_80, READ_20], tuple(SCORES), ("a", "i", "d", "-"), strict=bool(tie))
        if (t, sites) != (tp, sp):
            raise SystemExit("the C oracle and the Python twin disagree on MapRef / tie %d" % tie)
        maps.append({"name": "MapRef-REFx5-%s" % ("strict" if tie else "serial"), "ref": ref, "reads": [READ_80, READ_20], "scores": SCORES,
                     "tie_mode": tie, "total": t, "match_sites": [[b, r, q] for (b, (r, q)) in sites]})
    return {"_comment": "EngineerData-shaped goldens (src/metrics/EngineerData.java:23,26,29,51-224): REF x 5 against READ_80 and READ_20, "
                        "both tie modes; produced by oracle/sw_oracle.c and required equal from oracle/sw_oracle_py.py by "
                        "tools/gen_engineerdata_golden.py.  The reference ships no expected outputs: parity unpinned.",
            "kats": cases, "map_refs": maps}


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "engineerdata_small.json")
    with open(out, "w") as f:
        json.dump(build(), f, indent=1)
        f.write("\n")
    print("wrote", out)
