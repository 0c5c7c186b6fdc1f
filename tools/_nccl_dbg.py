import os, sys, faulthandler
faulthandler.enable()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
print("init...", flush=True)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("group up", flush=True)
t = torch.ones(1024, device="cuda")
dist.all_reduce(t); torch.cuda.synchronize(); print("allreduce ok", float(t.sum()), flush=True)
dist.broadcast(t, src=0); torch.cuda.synchronize(); print("broadcast ok", flush=True)
dist.destroy_process_group(); print("done", flush=True)
