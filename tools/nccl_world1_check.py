import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
import torch, torch.distributed as dist, numpy as np
from sparksmithwaterman_amd import distributed as swd
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
dev=torch.device("cuda",0)
r=swd.MaxReducer(dev); r.world=1
print(r([1,9,9],[10,11,12]))
# force the collective path with world 1
r.world=2-1
r2=swd.MaxReducer(dev)
r2.world=1
# exercise buffers + all_gather_into_tensor on a 1-rank group
r2.pay_np[:]=0; r2.d_pay.copy_(r2.h_pay, non_blocking=True)
dist.all_gather_into_tensor(r2.d_all, r2.d_pay)
r2.h_all.copy_(r2.d_all, non_blocking=True); torch.cuda.current_stream().synchronize()
t=time.perf_counter()
for _ in range(100):
    r2.d_pay.copy_(r2.h_pay, non_blocking=True); dist.all_gather_into_tensor(r2.d_all, r2.d_pay); r2.h_all.copy_(r2.d_all, non_blocking=True); torch.cuda.current_stream().synchronize()
print("us per reduce (1-rank nccl):", (time.perf_counter()-t)*1e4)
print(swd.global_max_with_ties([1,9,9],[10,11,12], device=dev))
dist.destroy_process_group()
