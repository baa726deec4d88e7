"""One long pair (default 8000 x 8000) through every pipeline, checked against the oracle (GPU box).  Test helper for
the LDS budget of the traceback kernels; not part of the product."""
import os, sys, time, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparksmithwaterman_amd as sw
from oracle import sw_oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n
rng = random.Random(7)
ref = "".join(rng.choice("ACGT") for _ in range(n))
# the read is a mutated copy of a slice of the reference, so the alignment is long
read = list(ref[: m])
for k in range(0, m, 17):
    read[k] = rng.choice("ACGT")
read = "".join(read)
es, ea = orc.opt_alignments((ref, read), (5, -3, -4), b"aid-", 0)
print("oracle score", es, "alignments", len(ea), "path", len(ea[0][1][0]) if ea else 0)
for mode in (1, 2, 0):
    c = sw.Context(0)
    c.set_option("mode", mode)
    try:
        t = time.perf_counter()
        b = c.upload([ref], [read]).run(sw.make_params())
        dt = time.perf_counter() - t
        ok = b.score(0) == es and b.alignments(0) == ea
        print("mode", mode, "ok" if ok else "MISMATCH", "%.1f ms" % (dt * 1e3))
        b.free()
    except Exception as e:
        print("mode", mode, "error:", e)
    c.close()
