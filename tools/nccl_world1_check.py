"""Rehearsal of MaxReducer's RCCL path on a ONE-rank nccl group (run on the GPU box): the blocking call, and the
pipelined submit/collect form bench.py uses, each timed.  Real multi-GPU numbers can only come from an 8-GPU node."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from sparksmithwaterman_amd import distributed as swd
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
r = swd.MaxReducer(dev, always_exchange=True)
assert r([1, 9, 9], [10, 11, 12]) == (9, [11, 12])
t = time.perf_counter()
for _ in range(200):
    r([1, 9, 9], [10, 11, 12])
print("us per blocking reduce (1-rank nccl): %.1f" % ((time.perf_counter() - t) / 200 * 1e6))
t = time.perf_counter()
prev = None
for _ in range(200):
    cur = r.submit([1, 9, 9], [10, 11, 12])
    if prev is not None:
        assert r.collect(prev) == (9, [11, 12])
    prev = cur
assert r.collect(prev) == (9, [11, 12])
print("us per pipelined reduce, host side (1-rank nccl): %.1f" % ((time.perf_counter() - t) / 200 * 1e6))
print(swd.global_max_with_ties([1, 9, 9], [10, 11, 12], device=dev))
dist.destroy_process_group()
