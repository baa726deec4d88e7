"""The drop-in boundary beyond symbol checks: the JNI shim's call sequence as a plain C99 program (the build image has
no JDK, so bindings/jni/swmi_jni.c itself cannot be compiled; its logic lives in bindings/jni/swmi_shim.c, which is),
re-entrancy of the C ABI (MapRef.call runs on every executor thread, src/sw/Distribution.java:32,403), and the
error / fallback paths of the runtime."""
import os
import random
import subprocess
import threading

import pytest

import sparksmithwaterman_amd as sw
from sparksmithwaterman_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_shim(tmp_path):
    exe = tmp_path / "shim_kat"
    lib = os.path.join(ROOT, "sparksmithwaterman_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "bindings", "jni"),
                           os.path.join(ROOT, "tests", "c", "shim_kat.c"), os.path.join(ROOT, "bindings", "jni", "swmi_shim.c"),
                           "-L", lib, "-lswmi", "-Wl,-rpath," + lib, "-o", str(exe)])
    return exe


def test_shim_sequence_is_c99_clean_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build_shim(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 3 and "no CPU fallback" in p.stdout


def test_jni_translation_unit_includes_what_it_uses():
    src = open(os.path.join(ROOT, "bindings", "jni", "swmi_jni.c")).read()
    assert "#include <stdio.h>" in src                      # snprintf (it had never met a compiler)
    assert "GetDirectBufferCapacity" in src and "GetArrayLength" in src


@pytest.mark.gpu
def test_shim_sequence_matches_kat3(tmp_path, kats):
    exe = _build_shim(tmp_path)
    for arg, name in (("serial", "KAT-3-serial"), ("strict", "KAT-3-strict")):
        k = next(x for x in kats if x["name"] == name)
        lines = subprocess.run([str(exe), arg], capture_output=True, text=True, check=True).stdout.splitlines()
        want = k.get("map_ref_sorted", k["alignments"])
        # line 1: the per-site call sequence; line 2: the same sites through the bulk accessor (swmi_ref_sites_packed)
        assert len(lines) == 2
        for out in (ln.split() for ln in lines):
            assert out[0] == str(k["score"]) and out[1] == str(len(want))
            assert out[2:] == ["%d:%s/%s" % (b, r, q) for b, r, q in want]


@pytest.mark.gpu
def test_ref_sites_packed_is_map_ref_of_a_whole_partition():
    """swmi_ref_sites_packed (what GpuSmithWaterman.MapPartition binds): totals, degenerate-site counts and every real match
    site of a range of references, equal to the per-site accessors and to the oracle's MapRef (Distribution.java:403-436)."""
    from oracle import sw_oracle as orc
    refs = ["CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGG" * 3, "ACGTTTGACCAacgtGGAC", "GGGG", "acgtACGTTGCAtgca" * 4, ""]
    reads = ["CATCTGACCAGGGCAGGCCTGG", "TTTT", "ACGTTGCA", "GGAC"]
    for zero_copy, strings in ((1, 1), (0, 1), (1, 0)):
        ctx = sw.Context(0)
        ctx.set_option("zero_copy", zero_copy)
        ctx.set_option("device_strings", strings)
        b = ctx.upload(refs, reads).run()
        packed = b.ref_sites_packed()
        assert len(packed) == len(refs)
        for r, (total, n_deg, sites) in enumerate(packed):
            wt, (_, ws) = orc.map_ref((">gi|r%d" % r, refs[r]), reads)
            assert total == wt == b.ref_total(r)
            assert [(0, ("", ""))] * n_deg + sites == ws == b.ref_match_sites(r)
        assert b.ref_sites_packed(1, 3) == packed[1:3] and b.ref_sites_packed(2, 2) == []
        with pytest.raises(sw.SwmiError):
            b.ref_sites_packed(3, 9)
        b.free()
        ctx.close()


@pytest.mark.gpu
def test_two_contexts_on_two_threads():
    """Two contexts, two host threads, different batches running at the same time (ctypes drops the GIL in the calls):
    each thread gets its own results, identical to what the same batch gives alone."""
    from oracle import sw_oracle as orc
    jobs = [synth.config_1k(n_refs=300, ref_len=700, read_len=150, seed=11),
            synth.config_ncbi(250, read_len=100, seed=12)]
    want = []
    for refs, reads in jobs:
        r = orc.bench(refs, reads, nthreads=8, per_pair=True)
        want.append((r["pair_score"], r["pair_naln"]))
    errors, got = [], [None, None]

    def worker(k):
        try:
            ctx = sw.Context(0)
            refs, reads = jobs[k]
            b = ctx.upload(refs, reads)
            for _ in range(6):
                b.run()
                sc, na = b.pair_results()
                assert list(sc) == want[k][0] and [int(x) for x in na] == want[k][1]
            # a third, small batch per thread through the one-shot mirror class
            s, a = sw.SmithWaterman.OptAlignments(ctx).call([refs[0], reads[0]])
            assert (s, a) == orc.opt_alignments((refs[0], reads[0]))
            got[k] = True
            b.free()
            ctx.close()
        except BaseException as e:      # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors
    assert got == [True, True]


@pytest.mark.gpu
def test_two_batches_in_flight_alternating():
    """bench.py's step loop: two batches in flight on one GPU, one context (stream, host thread, result block) each, started
    alternately with swmi_batch_run_async so that one batch's sweep runs beside the other's traceback.  Different data in
    the two batches; every step's results -- scores, counts, every alignment string of sampled pairs -- against the oracle."""
    from oracle import sw_oracle as orc
    jobs = [synth.config_1k(n_refs=400, ref_len=900, read_len=150, seed=21),
            synth.config_1k(n_refs=350, ref_len=1200, read_len=120, seed=22)]
    want = []
    for refs, reads in jobs:
        r = orc.bench(refs, reads, nthreads=8, per_pair=True)
        want.append((r["pair_score"], r["pair_naln"]))
    ctxs = [sw.Context(0), sw.Context(0)]
    bs = [c.upload(*job) for c, job in zip(ctxs, jobs)]
    in_flight = [False, False]

    def retire(i):
        bs[i].wait()
        sc, na = bs[i].pair_results()
        assert [int(x) for x in sc] == want[i][0] and [int(x) for x in na] == want[i][1]
        refs, reads = jobs[i]
        for p in (0, 7, len(refs) - 1):
            assert bs[i].alignments(p) == orc.opt_alignments((refs[p], reads[0]))[1]

    for k in range(12):
        i = k % 2
        if in_flight[i]:
            retire(i)
        bs[i].run_async()
        in_flight[i] = True
    for i in range(2):
        retire(i)
    for b in bs:
        b.free()
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_repeated_runs_header_ring_and_deferred_record_stream():
    """Host-side bookkeeping of repeated runs (swmi_api.cpp): a launch of whole-pair kernels only takes its arena header from a
    ring of 1024 zeroed slots (1100 runs wrap it); the record stream of a single-launch run stays in the pinned block until an
    alignment is asked for, so a second run must not disturb what the first one handed out, and results read after any number
    of runs are the last run's."""
    from oracle import sw_oracle as orc
    ref = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"      # EngineerData.java:23
    refs = [ref * 5] * 300 + [ref[::-1] * 5] * 20
    reads = [ref[7:87], "ACTGACTGACTGACTGACTG"]
    want = {(r, q): orc.opt_alignments((r, q)) for r in set(refs) for q in reads}
    for resident, tfused in ((1, 0), (0, 1), (0, 0)):
        ctx = sw.Context(0)
        ctx.set_option("resident", resident)
        ctx.set_option("tfused", tfused)
        b = ctx.upload(refs, reads)
        p1 = sw.make_params((5, -3, -4))
        p2 = sw.make_params((1, -1, -1))
        n_runs = 1100 if resident else 40
        for k in range(n_runs):
            b.run(p1)
            if k % 275 == 3:
                for pair in (0, 1, 2 * 300, 2 * 319 + 1):
                    es, ea = want[(refs[pair // 2], reads[pair % 2])]
                    assert b.score(pair) == es and b.alignments(pair) == ea, (resident, tfused, k, pair)
        first = b.alignments(5)                  # (indexes the stream of the last run)
        b.run(p2)                                # another run with other scores: its stream replaces the pinned block's content
        es2, ea2 = orc.opt_alignments((refs[5 // 2], reads[5 % 2]), (1, -1, -1))
        assert b.score(5) == es2 and b.alignments(5) == ea2
        b.run(p1)
        assert b.alignments(5) == first == want[(refs[2], reads[1])][1]
        b.free()
        ctx.close()


@pytest.mark.gpu
def test_strip_pipeline_give_up_falls_back_to_one_wave_sweep():
    """Long reads are swept one wavefront per strip, each strip waiting for the one above it.  With the items dispatched
    consumer-first and a spin budget of one poll the consumers give up; the runtime must then re-run the chunk with the
    one-wavefront sweep and still return the oracle's results (the batch used to fail with SWMI_ERR_HIP)."""
    from oracle import sw_oracle as orc
    rng = random.Random(5)
    refs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (1500, 2100)]
    reads = [refs[0][100:900], refs[1][50:600], refs[0][300:450]]
    ctx = sw.Context(0)
    ctx.set_option("mode", 1)
    ctx.set_option("debug_reverse_strips", 1)
    ctx.set_option("debug_strip_spins", 1)
    b = ctx.upload(refs, reads).run()
    assert b.timing().strip_fallbacks >= 1
    for r, ref in enumerate(refs):
        for q, read in enumerate(reads):
            es, ea = orc.opt_alignments((ref, read))
            assert b.score(r * len(reads) + q) == es
            assert b.alignments(r * len(reads) + q) == ea
    b.free()
    # the same with every pair's sweep cut into column chunks (each chunk a strip pipeline of its own, swmi_device.h: StripItem)
    ctx.set_option("col_chunks", 3)
    b = ctx.upload(refs, reads).run()
    assert b.timing().strip_fallbacks >= 1
    assert [b.score(p) for p in range(6)] == [orc.opt_alignments((ref, read))[0] for ref in refs for read in reads]
    assert b.alignments(0) == orc.opt_alignments((refs[0], reads[0]))[1]
    b.free()
    ctx.set_option("col_chunks", 0)
    # the knobs off again: the pipeline itself, no fallback
    ctx.set_option("debug_reverse_strips", 0)
    ctx.set_option("debug_strip_spins", 0)
    b = ctx.upload(refs, reads).run()
    assert b.timing().strip_fallbacks == 0
    assert b.score(0) == orc.opt_alignments((refs[0], reads[0]))[0]
    b.free()
    ctx.close()


@pytest.mark.gpu
def test_scores_only_is_the_sweep_alone():
    """option scores_only: every pair's score and MapRef's totals from the sweep kernels alone -- equal to the full path's, for
    every sweep shape (one wavefront, column chunks, strips of a long read, a byte alphabet, degenerate pairs) -- and the accessors
    of what was not computed fail instead of inventing it.  Streams carry the option to their chunks."""
    import numpy as np
    import random
    rng = random.Random(17)
    refs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (30, 400, 2100, 5000)] + ["GGGG", "acgtnACGTNxx" * 20]
    reads = ["".join(rng.choice("ACGT") for _ in range(m)) for m in (20, 150, 300)] + ["TTTT", refs[2][100:240].lower()]
    ctx = sw.Context(0)
    full = ctx.upload(refs, reads).run()
    want_scores, _ = full.pair_results()
    want_totals = full.ref_totals()
    ctx.set_option("scores_only", 1)
    b = ctx.upload(refs, reads).run()
    assert list(b.scores()) == list(want_scores) and list(b.ref_totals()) == list(want_totals)
    assert b.score(7) == int(want_scores[7])
    deg = [p for p in range(len(refs) * len(reads)) if want_scores[p] == 0]
    assert deg and b.n_alignments(deg[0]) == full.n_alignments(deg[0])           # (a maximum of 0: m * n, known without a traceback)
    live = int(np.argmax(want_scores))
    for call in (lambda: b.n_alignments(live), lambda: b.alignments(live), lambda: b.pair_results(), lambda: b.ref_match_sites(0),
                 lambda: b.ref_sites_packed()):
        with pytest.raises(sw.SwmiError):
            call()
    st = ctx.stream(reads, slots=2, chunk_bytes=64 << 10).push(refs).finish()
    assert list(st.totals()) == list(want_totals)
    st.close()
    ctx.set_option("scores_only", 0)
    again = ctx.upload(refs, reads).run()
    assert again.alignments(live) == full.alignments(live)
    for x in (full, b, again):
        x.free()
    ctx.close()
