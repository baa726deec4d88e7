"""configs[3] in small: ONE reference set sharded over two rank processes, reduced over torch.distributed, checked against
the oracle (Distribution.java:337-353, 600-613).  Both ranks use GPU 0 (a gpurun box has one card) and gloo for the
exchange; on a node with two GPUs the same module runs one rank per GPU over RCCL (backend "nccl")."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))


@pytest.mark.gpu
def test_two_ranks_shard_one_reference_set(tmp_path):
    from sparksmithwaterman_amd import synth, distributed as swd
    from oracle import sw_oracle as orc
    n_refs, n_reads, k = 2000, 32, 8
    env = dict(os.environ, SWMI_ONE_GPU="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    # the launcher is a fresh child that never touches the GPU; it starts the two ranks (children, never a re-exec)
    rc = subprocess.run([sys.executable, "-m", "sparksmithwaterman_amd.sharded", "--n-refs", str(n_refs), "--n-reads", str(n_reads),
                         "--world", "2", "--top-k", str(k), "--out", str(tmp_path)], cwd=ROOT, env=env, timeout=900)
    assert rc.returncode == 0
    ranks = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [r["backend"] for r in ranks] == ["gloo", "gloo"]

    refs, reads = synth.config_multi_read(n_refs, n_reads, seed=3)
    o = orc.bench(refs, reads, nthreads=_cores(), per_pair=True)
    sc = np.asarray(o["pair_score"], dtype=np.int64).reshape(n_refs, n_reads)
    totals = sc.sum(axis=1)                                  # MapRef's total (Distribution.java:424); no wrap at these sizes
    # every reference is on exactly one rank, and its total is the oracle's
    ids = np.concatenate([r["local_ids"] for r in ranks])
    assert sorted(ids.tolist()) == list(range(n_refs))
    lens = np.array([len(r) for r in refs])
    for r in ranks:
        assert r["local_ids"] == swd.shard_by_length(lens, r["rank"], 2).tolist()
        assert r["local_totals"] == totals[r["local_ids"]].tolist()
    # length-balanced: the two shards hold the same number of references and nearly the same number of cells
    c0, c1 = ranks[0]["cells"], ranks[1]["cells"]
    assert len(ranks[0]["local_ids"]) == len(ranks[1]["local_ids"]) and abs(c0 - c1) < 0.01 * (c0 + c1)
    # the driver's reduce: max with ties (control semantics, `int max = 0`) and top-K, identical on both ranks
    best = max(int(totals.max()), 0)
    winners = np.flatnonzero(totals == best).tolist()
    order = sorted(range(n_refs), key=lambda i: (-int(totals[i]), i))[:k]
    for r in ranks:
        assert r["best"] == best and r["winners"] == winners
        assert [tuple(x) for x in r["top_k"]] == [(int(totals[i]), i) for i in order]
