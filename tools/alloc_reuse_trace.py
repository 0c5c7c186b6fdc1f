#!/usr/bin/env python3
"""Which device-memory blocks does the caching allocator hand to tensors of DIFFERENT streams during the pass the launch tapes are recorded from?  In the eager pass such a
re-use is ordered by the allocator itself (host-side event queries); a replayed tape only has the dependencies it recorded, so every cross-stream re-use is a place where a
replay can differ from the recording.  Prints the cross-stream re-uses (address, size, the two streams, the allocating frames)."""
import os, sys, types, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False)
torch.cuda.memory._record_memory_history(enabled="all", context="alloc", stacks="python", max_entries=400000)
eng.step(x, lab)                     # warm-up passes + the recorded pass
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
tr = [e for t in snap["device_traces"] for e in t]
print(len(tr), "allocator events")
last = {}          # addr -> (stream, size, frames) of the latest alloc
pairs = collections.Counter()
examples = {}
def top(frames):
    fs = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in frames if "veloxseg_amd" in f.get("filename", "") or "engine" in f.get("filename", "")]
    return " < ".join(fs[:3]) if fs else (f"{os.path.basename(frames[0]['filename'])}:{frames[0]['line']}" if frames else "?")
for e in tr:
    if e["action"] != "alloc":
        continue
    a, s, n = e["addr"], e["stream"], e["size"]
    fr = top(e.get("frames", []))
    for (a0, (s0, n0, fr0)) in list(last.items()) if False else ():
        pass
    if a in last and last[a][0] != s:
        key = (last[a][2], fr, last[a][0], s)
        pairs[key] += 1
        examples.setdefault(key, (a, last[a][1], n))
    last[a] = (s, n, fr)
print(len(pairs), "distinct (previous owner -> new owner) cross-stream re-uses of the same block address")
for (f0, f1, s0, s1), c in pairs.most_common(40):
    a, n0, n1 = examples[(f0, f1, s0, s1)]
    print(f"{c:4d}x  {n0:>10d} B (stream {s0:#x}) {f0}\n        -> {n1:>10d} B (stream {s1:#x}) {f1}")
