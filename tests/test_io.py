"""Native sequence-file reader and result writer against the Python restatement of InOutOps (CPU only),
and the file-level drivers end to end on the GPU."""
import os

import pytest

import sparksmithwaterman_amd as sw
from sparksmithwaterman_amd import io as swio
from oracle import io_oracle_py as ioo

REF = "CCTGGGTCCTGCCTCGCATCTGACCAGGGCAGGTGGCCTCCTCATCACACTGCTGCCTCTGCTGTTGGCCCTGCTCATGA"   # EngineerData.java:23
READ_20 = "ACTGACTGACTGACTGACTG"                                                            # EngineerData.java:29

CASES = {
    "plain.fa": ">gi|1 first\nACGTACGT\nACGT\n>gi|2 second\nGGGGCCCC\n",
    "crlf.fa": ">gi|1\r\nACGT\r\nAC GT \r\n>gi|2\r\n\r\nTT\r\n",
    "noeol.fa": ">gi|a\nAC\n>gi|b\nGT",
    "untrimmed.fa": ">gi|x meta with spaces  \n  ACGT  \n\tAC\n>gi|y\n>gi|z\nA\n",
    "lonecr.fa": ">gi|1\rACGT\rAC\r",
    "reads_meta.txt": ">gi|reads file\nACGT\n\n  AC GT  \n>gi|not skipped\nTTTT",
    "reads_nometa.txt": "ACGTAC\nGG\n",
    "reads_oneline.txt": ">gi only metadata",
}


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("seqfiles")
    for name, text in CASES.items():
        with open(d / name, "w", newline="") as f:
            f.write(text)
    return d


def test_io_symbols_exported():
    lib = sw._capi.load()
    for name, _, _ in swio.IO_SYMBOLS:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("name", [n for n in CASES if n.endswith(".fa")])
def test_get_ref_seqs_matches_restatement(files, name):
    got = swio.InOutOps.GetRefSeqs().call(files / name, ">gi")
    assert got == ioo.get_ref_seqs(files / name, ">gi")


@pytest.mark.parametrize("name", list(CASES))
def test_get_reads_matches_restatement(files, name):
    got = swio.InOutOps.GetReads().call(files / name, ">gi")
    assert got == ioo.get_reads(files / name, ">gi")


def test_reads_quirks_spelled_out(files):
    # every line after the first is a read: blank lines and later '>' lines included (InOutOps.java:75-76)
    assert swio.InOutOps.GetReads().call(files / "reads_meta.txt", ">gi") == ["ACGT", "", "AC GT", ">gi|not skipped", "TTTT"]
    assert swio.InOutOps.GetReads().call(files / "reads_oneline.txt", ">gi") == []
    # reference lines are appended untrimmed (:148); an empty record is kept
    assert swio.InOutOps.GetRefSeqs().call(files / "untrimmed.fa", ">gi") == [
        [">gi|x meta with spaces  ", "  ACGT  \tAC"], [">gi|y", ""], [">gi|z", "A"]]


def test_reader_errors(files, tmp_path):
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    with pytest.raises(sw.SwmiError):
        swio.InOutOps.GetReads().call(empty, ">gi")
    with pytest.raises(sw.SwmiError):
        swio.InOutOps.GetRefSeqs().call(empty, ">gi")
    with pytest.raises(sw.SwmiError):
        swio.InOutOps.GetRefSeqs().call(files / "reads_nometa.txt", ">gi")     # no leading metadata line
    with pytest.raises(sw.SwmiError):
        swio.InOutOps.GetReads().call(tmp_path / "missing.txt", ">gi")


def test_packed_reader_feeds_offsets(files):
    s = swio.read_refs_packed(files / "plain.fa", ">gi")
    assert len(s) == 2 and s.offsets == [0, 12, 20] and s.blob == b"ACGTACGTACGTGGGGCCCC"
    assert s.metadata == [">gi|1 first", ">gi|2 second"]


def test_output_str_matches_restatement():
    reads = ["ACGT", ""]
    opt = [([">gi|b", "ACGTT"], [(1, ("ACGT", "ACGT")), (3, ("G_T", "GAT"))]), ([">gi|a", "AC"], [])]
    got = swio.InOutOps.GetOutputStr().call(reads, ((7, 2), 20, 123), opt)
    assert got == ioo.get_output_str(reads, (7, 2), 20, 123, opt)
    nl = os.linesep
    assert got.startswith("Execution Time = 123 ms" + nl + nl + "# Reference Sequences = 7" + nl + "# Reads = 2" + nl + nl + "Input:" + nl)
    assert ("Maximum alignment score = 20" + nl + "Reference:" + nl + ">gi|b" + nl + "ACGTT" + nl + nl + "\tIndex = 1" + nl) in got


def test_directory_crawler_depth_first(tmp_path):
    (tmp_path / "b").mkdir()
    (tmp_path / "b" / "z.txt").write_text("x")
    (tmp_path / "a.txt").write_text("x")
    (tmp_path / "c.txt").write_text("x")
    c = swio.DirectoryCrawler(str(tmp_path))
    seen = []
    while c.hasNext():
        seen.append(os.path.relpath(c.next(), tmp_path))
    assert seen == ["a.txt", os.path.join("b", "z.txt"), "c.txt"]
    with pytest.raises(FileNotFoundError):
        swio.DirectoryCrawler(str(tmp_path / "nope"))


@pytest.mark.gpu
def test_drivers_write_the_control_drivers_result_files(tmp_path):
    ref_dir, in_dir = tmp_path / "reference", tmp_path / "input"
    out_gpu, out_cpu = tmp_path / "out_gpu", tmp_path / "out_cpu"
    for d in (ref_dir, in_dir, out_gpu, out_cpu, ref_dir / "sub"):
        d.mkdir()
    # EngineerData-shaped files: references are REF repeated (EngineerData.java:118,139), 80 characters per line
    def fasta(recs):
        out = []
        for meta, seq in recs:
            out.append(meta)
            out.extend(seq[k:k + 80] for k in range(0, len(seq), 80))
        return "\n".join(out) + "\n"
    (ref_dir / "ref1.fa").write_text(fasta([(">gi|ref1", REF * 3), (">gi|ref0", REF[::-1] * 2), (">gi|ref2", REF * 3)]))
    (ref_dir / "sub" / "ref9.fa").write_text(fasta([(">gi|ref9", REF[5:70] + "ACGT" * 9)]))
    (in_dir / "input1.txt").write_text(">gi reads\n" + REF[10:50] + "\n" + READ_20 + "\n")
    (in_dir / "input2.txt").write_text("TTTTTTTTGGGGG\n")
    ctx = sw.Context(0)
    io_args = [str(ref_dir), str(in_dir), ">gi", str(out_gpu), None, None]
    assert sw.Distribution.DistributeReference(ctx).call(io_args, None) is None
    expect = ioo.no_distribution(str(ref_dir), str(in_dir), ">gi", str(out_cpu))
    for k, text in enumerate(expect, 1):
        got = open(out_gpu / ("result%d.txt" % k), newline="").read()
        # identical apart from the wall-clock line (InOutOps.java:249)
        assert got.split(os.linesep, 1)[1] == text.split(os.linesep, 1)[1]
        assert got.startswith("Execution Time = ")
    out2 = tmp_path / "out_ctl"
    out2.mkdir()
    sw.Distribution.NoDistribution(ctx).call([str(ref_dir), str(in_dir), ">gi", str(out2), "res", ".out"], ([5, -3, -4], ["a", "i", "d", "-"]))
    assert open(out2 / "res1.out", newline="").read().split(os.linesep, 1)[1] == expect[0].split(os.linesep, 1)[1]
    ctx.close()


def _write_fasta(path, refs, width=80):
    with open(path, "w") as f:
        for k, r in enumerate(refs):
            f.write(">gi|ref%d\n" % k)
            for x in range(0, len(r), width):
                f.write(r[x:x + width] + "\n")


@pytest.mark.gpu
def test_stream_from_memory_matches_one_batch(tmp_path):
    """swmi_stream_push: the references cut into many small chunks over two slots give, chunk by chunk, what one batch
    gives -- totals, every pair's score and alignment count, and the alignment strings (from the bytes the stream kept)."""
    import numpy as np
    from sparksmithwaterman_amd import synth
    from oracle import sw_oracle as orc
    refs, reads = synth.config_ncbi(1500, read_len=150, seed=21)
    reads = reads + [refs[7][30:130].decode()]
    ctx = sw.Context(0)
    b = ctx.upload(refs, reads).run()
    want_sc, want_na = b.pair_results()
    st = ctx.stream(reads, slots=2, chunk_bytes=200 << 10)
    st.push(refs[:900]).push(refs[900:]).finish()
    assert st.n_refs() == len(refs)
    chunks = st.chunks()
    assert len(chunks) > 8 and chunks[0][0] == 0
    assert list(st.totals()) == list(b.ref_totals())
    got_sc = np.concatenate([c.pair_results()[0] for _, c in chunks])
    got_na = np.concatenate([c.pair_results()[1] for _, c in chunks])
    assert list(got_sc) == list(want_sc) and list(got_na) == list(want_na)
    for first, c in chunks[::3]:
        r = first + c.n_refs - 1
        assert c.ref_match_sites(c.n_refs - 1) == orc.map_ref((">gi|x", refs[r]), reads)[1][1]
    st.close()
    b.free()
    ctx.close()


@pytest.mark.gpu
def test_stream_views_keep_their_stream_alive_and_die_with_it():
    """`ctx.stream(reads).push(refs).finish().chunks()` leaves no name for the Stream -- the views must keep it
    alive (swmi_stream_close frees every result batch); after an explicit close a view raises instead of reading freed memory."""
    import gc
    from sparksmithwaterman_amd import synth
    from oracle import sw_oracle as orc
    refs, reads = synth.config_ncbi(40, read_len=150, seed=5)
    ctx = sw.Context(0)
    chunks = ctx.stream(reads, slots=2, chunk_bytes=64 << 10).push(refs).finish().chunks()
    gc.collect()
    first, c = chunks[-1]
    assert c.score(0) == orc.opt_alignments((refs[first], reads[0]))[0]
    assert c.alignments(0) == orc.opt_alignments((refs[first], reads[0]))[1]
    st = c._owner
    st.close()
    with pytest.raises(sw.SwmiError):
        c.score(0)
    with pytest.raises(sw.SwmiError):
        chunks[0][1].alignments(0)
    ctx.close()


@pytest.mark.gpu
def test_stream_from_fasta_file_100k_references(tmp_path):
    """configs[2] shape through the streamed path: a FASTA file of 100,000 NCBI-shaped references (log-normal
    lengths, median 1,609 bp), parsed segment-wise into pinned memory, canonicalised on the GPU, aligned chunk by chunk on
    three overlapping slots.  Checked against the oracle's multi-threaded pass over the same 100,000 pairs: sum of scores, sum of
    alignment counts, the winner and its alignments (whose bytes are re-read from the mapped file)."""
    import numpy as np
    from sparksmithwaterman_amd import synth
    from oracle import sw_oracle as orc
    refs, reads = synth.config_ncbi(100000, read_len=150, seed=2)
    path = tmp_path / "refs.fa"
    _write_fasta(path, [r.decode() for r in refs])
    ctx = sw.Context(0)
    st = ctx.stream(reads, slots=3, chunk_bytes=16 << 20)
    st.push_file(path, ">gi", 6).finish()
    assert st.n_refs() == len(refs)
    totals = st.totals()
    n_aln = sum(int(c.pair_results()[1].sum()) for _, c in st.chunks())
    ob = orc.bench(refs, reads, nthreads=16)
    assert int(totals.astype(np.int64).sum()) == ob["sum_score"]
    assert n_aln == ob["sum_aln"]
    w = int(totals.argmax())
    assert st.metadata(w) == ">gi|ref%d" % w
    es, ea = orc.opt_alignments((refs[w], reads[0]))
    for first, c in st.chunks():
        if first <= w < first + c.n_refs:
            assert c.score(w - first) == es and c.alignments(w - first) == ea
    assert st.stats().chunks == len(st.chunks()) >= 10
    st.close()
    ctx.close()


@pytest.mark.gpu
def test_stream_errors_and_oversized_records(tmp_path):
    """The stream's error paths (a missing file, a file without a leading metadata line: the reference dies with a
    NullPointerException there, InOutOps.java:148) and a record longer than a chunk (its pinned buffer grows)."""
    import numpy as np
    from sparksmithwaterman_amd import synth
    ctx = sw.Context(0)
    reads = ["ACGTACGTTGCAACGTTGCA"]
    st = ctx.stream(reads, slots=2, chunk_bytes=64 << 10)
    with pytest.raises(sw.SwmiError):
        st.push_file(tmp_path / "missing.fa")
    st.close()
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n>gi|late\nACGT\n")
    st = ctx.stream(reads, slots=2, chunk_bytes=64 << 10)
    with pytest.raises(sw.SwmiError):
        st.push_file(bad)
    st.close()
    # one 300 kbp record between short ones, chunks of 64 KiB
    rng = synth.SplitMix64(77)
    refs = [rng.bases(500), rng.bases(300000), rng.bases(700), rng.bases(70000)]
    path = tmp_path / "big.fa"
    _write_fasta(path, [r.decode() for r in refs])
    st = ctx.stream(reads, slots=2, chunk_bytes=64 << 10)
    st.push_file(path).finish()
    b = ctx.upload(refs, reads).run()
    assert list(st.totals()) == list(b.ref_totals())
    assert st.n_refs() == 4 and len(st.chunks()) >= 2
    assert [st.metadata(k) for k in range(4)] == [">gi|ref%d" % k for k in range(4)]
    first, c = st.chunks()[-1]
    assert c.alignments(c.n_refs * 1 - 1) == b.alignments(3)
    st.close()
    st = ctx.stream(reads, slots=1, chunk_bytes=64 << 10)
    st.push(refs).finish()
    assert list(st.totals()) == list(b.ref_totals())
    st.close()
    b.free()
    ctx.close()
