#!/usr/bin/env python3
"""RCCL on the one GPU there is: a torch.distributed "nccl" process group of ONE rank; TrainEngine(force_comm=True) issues its two all-reduce buckets
through ProcessGroupNCCL's own stream and events for real.  A one-rank all-reduce moves no bytes between GPUs, so this measures exactly the unknown a one-GPU box can
measure: what a process-group-owned stream costs beside the tape's four busy hardware queues, per placement (VELOXSEG_COMM_PLACEMENT).  Prints one JSON object.
argv: [workload] [batch] [steps]"""
import json, os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda:0")
res = {"workload": wl, "batch": B, "backend": "nccl (RCCL), world size 1", "steps": steps, "placements": {}}
for placement in ("lane", "fresh_after", "fresh_before"):
    os.environ["VELOXSEG_COMM_PLACEMENT"] = placement
    torch.manual_seed(12345)
    model = VeloxSeg(**cfg).to(dev)
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, dev, num_modal=len(cfg["in_ch"]))
    eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=True, force_comm=True, pipeline_tail=os.environ.get("VX_PIPE", "1") == "1")
    x, lab = synth(cfg, B, dev, 12345)
    eng.step(x, lab)
    assert eng.use_graph and eng.dp and eng.overlap

    def timed(n):
        for _ in range(20):
            eng.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    ms = timed(steps)
    eng.skip_comm = True
    ms_nc = timed(steps)
    eng.skip_comm = False
    eng.comm_profile = []
    for _ in range(50):
        eng.step()
    rep = eng.comm_report()
    eng.comm_profile = None
    res["placements"][placement] = {"step_ms": round(ms, 3), "step_ms_no_comm": round(ms_nc, 3), "exposed_ms": round(ms - ms_nc, 3), "buckets": rep}
    del eng, model
    torch.cuda.synchronize()
dist.destroy_process_group()
out = os.environ.get("VX_OUT")
if out:
    open(out, "w").write(json.dumps(res, indent=1) + "\n")
print("RESULT " + json.dumps(res))
