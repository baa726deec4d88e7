#!/usr/bin/env python3
"""configs[2] end to end as BASELINE.md defines it: FASTA file bytes -> results in host memory, through the streamed path
(swmi_stream_push_file: segment-wise parse into pinned memory, raw H2D, encode on the GPU, sweep + traceback per chunk,
three chunk workers overlapping each other).

    python tools/stream_run.py --n-refs 1000000 [--oracle] > gpurun_out/stream_1M.json

The FASTA file is written in the reference's format (">gi|ref<k>" + 80-character lines, EngineerData.java:139) to
--dir (default /dev/shm so that the timed region reads page-cache-resident bytes, as a warm Spark input split would).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_fasta(path, refs, width=80):
    import numpy as np
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        buf = []
        size = 0
        for k, r in enumerate(refs):
            a = np.frombuffer(r, dtype=np.uint8)
            full = len(a) // width
            body = a[:full * width].reshape(full, width)
            lines = np.concatenate([body, np.full((full, 1), 10, dtype=np.uint8)], axis=1).tobytes()
            tail = a[full * width:].tobytes()
            buf.append(b">gi|ref%d\n" % k)
            buf.append(lines)
            if tail:
                buf.append(tail + b"\n")
            size += len(lines) + len(tail) + 16
            if size > (64 << 20):
                f.write(b"".join(buf))
                buf, size = [], 0
        f.write(b"".join(buf))
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-refs", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--chunk-mb", type=int, default=32)
    ap.add_argument("--parse-threads", type=int, default=6)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--oracle", action="store_true", help="check the checksums against the CPU oracle (about 1 s per 10k refs on 16 cores)")
    ap.add_argument("--monolithic", action="store_true", help="also time the one-batch path (upload + run)")
    args = ap.parse_args()

    import numpy as np
    import sparksmithwaterman_amd as sw
    from sparksmithwaterman_amd import synth

    g0 = time.perf_counter()
    refs, reads = synth.config_ncbi(args.n_refs, read_len=args.read_len, seed=2)
    gen_s = time.perf_counter() - g0
    path = os.path.join(args.dir, "swmi_stream_%d.fa" % args.n_refs)
    wr_s = write_fasta(path, refs)
    cells = sum(len(r) for r in refs) * len(reads[0])
    out = {"n_refs": args.n_refs, "cells": cells, "file_bytes": os.path.getsize(path), "generate_s": round(gen_s, 2),
           "write_fasta_s": round(wr_s, 2), "slots": args.slots, "chunk_mb": args.chunk_mb, "parse_threads": args.parse_threads,
           "runs": []}
    ctx = sw.Context(0)
    ctx.set_option("profiling", 1)
    try:
        for rep in range(args.reps):
            o0 = time.perf_counter()
            st = ctx.stream(reads, slots=args.slots, chunk_bytes=args.chunk_mb << 20)
            open_s = time.perf_counter() - o0
            t0 = time.perf_counter()
            st.push_file(path, ">gi", args.parse_threads)
            st.finish()
            e2e = time.perf_counter() - t0
            totals = st.totals()
            n_aln = 0
            for first, b in st.chunks():
                n_aln += int(b.pair_results()[1].sum())
            s = st.stats()
            out["runs"].append({
                "stream_open_s": round(open_s, 4), "end_to_end_s": round(e2e, 4),
                "gcups_end_to_end": round(cells / e2e / 1e9, 1),
                "gcups_incl_open": round(cells / (e2e + open_s) / 1e9, 1),
                "chunks": s.chunks, "parse_ms_sum": round(s.parse_ms, 1), "upload_ms_sum": round(s.upload_ms, 1),
                "run_ms_sum": round(s.run_ms, 1), "gpu_sweep_ms_sum": round(s.gpu_sweep_ms, 1),
                "gpu_traceback_ms_sum": round(s.gpu_traceback_ms, 1),
                "sum_totals": int(totals.astype(np.int64).sum()), "sum_alignments": n_aln,
                "winner": int(totals.argmax()), "winner_total": int(totals.max()), "winner_metadata": st.metadata(int(totals.argmax()))})
            if rep == args.reps - 1:
                w = int(totals.argmax())
                for first, b in st.chunks():
                    if first <= w < first + b.n_refs:
                        out["winner_alignments"] = b.ref_match_sites(w - first)[:2]
            st.close()
        if args.monolithic:
            t0 = time.perf_counter()
            b = ctx.upload(refs, reads)
            up = time.perf_counter() - t0
            b.run()
            t1 = time.perf_counter()
            b.run()
            run = time.perf_counter() - t1
            out["monolithic"] = {"upload_s": round(up, 3), "run_s": round(run, 4),
                                 "sum_totals": int(b.ref_totals().astype(np.int64).sum()),
                                 "gcups_incl_upload": round(cells / (up + run) / 1e9, 1)}
            b.free()
        if args.oracle:
            from oracle import sw_oracle as orc
            t0 = time.perf_counter()
            ob = orc.bench(refs, reads, nthreads=min(16, os.cpu_count() or 1))
            out["oracle"] = {"sum_score": ob["sum_score"], "sum_aln": ob["sum_aln"], "seconds": round(time.perf_counter() - t0, 1),
                             "match": ob["sum_score"] == out["runs"][-1]["sum_totals"] and ob["sum_aln"] == out["runs"][-1]["sum_alignments"]}
    finally:
        ctx.close()
        os.unlink(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
