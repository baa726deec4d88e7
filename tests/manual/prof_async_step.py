"""Times the sections of bench.py's multi-GPU step loop (run_async / reduce submit / collect / wait) on a one-rank RCCL
group.  Measured on one MI355X: async 1, submit 34, collect 3, wait 160, totals 6 us = 204 us per step; blocking run 188 us."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29555")
import numpy as np, torch, torch.distributed as dist
import sparksmithwaterman_amd as sw
from sparksmithwaterman_amd import synth, distributed as swd
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
refs, reads = synth.config_1k(1000, 2000, 150, seed=1)
ctx = sw.Context(0); batch = ctx.upload(refs, reads); params = sw.make_params()
red = swd.MaxReducer(dev, always_exchange=True); gids=np.arange(1000)
for _ in range(5): batch.run(params)
T={k:0.0 for k in ("async","submit","collect","wait","totals","total")}
prev=None; pend=[]
N=200
t00=time.perf_counter()
for _ in range(N):
    t0=time.perf_counter(); batch.run_async(params); t1=time.perf_counter()
    if prev is not None:
        pend.append(red.submit(prev,gids))
    t2=time.perf_counter()
    if len(pend)>1: red.collect(pend.pop(0))
    t3=time.perf_counter(); batch.wait(); t4=time.perf_counter()
    prev=batch.ref_totals(); t5=time.perf_counter()
    T["async"]+=t1-t0; T["submit"]+=t2-t1; T["collect"]+=t3-t2; T["wait"]+=t4-t3; T["totals"]+=t5-t4
T["total"]=time.perf_counter()-t00
print({k: round(v/N*1e6,1) for k,v in T.items()})
t00=time.perf_counter()
for _ in range(N): batch.run(params); prev=batch.ref_totals()
print("plain run us", round((time.perf_counter()-t00)/N*1e6,1))
dist.destroy_process_group()
