#!/usr/bin/env python3
"""Which device memory differs after a CONCURRENT replay of one stage tape, compared with the same tape launched node by node on one stream?  For a timing-dependent
deviation that no re-ordering on one stream reproduces (tools/tape_soak.py VX_SOAK_SERIAL / VX_SOAK_EXHAUSTIVE find nothing, VX_SOAK_HYBRID names the stage).
Every address a node of the stage passes to a kernel starts a region (up to the next such address of the same allocator segment); after each concurrent replay of the
stage every region is compared with its contents after the serial launch.  Prints the regions that differ, what differs, and the nodes that reference them.
argv: [workload] [batch] [tries] [stage]"""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd import _hip as H, functional as VF
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
from veloxseg_amd.tape_audit import RawMem, device_segments, node_pointers, tape_signatures, tape_layout

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
tries = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
stage = sys.argv[4] if len(sys.argv) > 4 else "enc_fwd"
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False)
eng.step(x, lab)
torch.cuda.synchronize()
assert eng.use_graph and eng.graphs is not None, "capture failed"
tape = eng.graphs[stage]
segs = device_segments()
sigs = tape_signatures(tape)
lane, waits, names, grid = tape_layout(tape)
refs = {}                       # address -> [(node, param, type, const)]
for i in range(tape.n_nodes):
    for k, ty, const, v in node_pointers(tape, i, sigs[i], segs):
        refs.setdefault(v, []).append((i, k, ty, const))
import bisect
bases = [s[0] for s in segs]
addrs = sorted(a for a in refs if bisect.bisect_right(bases, a) > 0 and a < segs[bisect.bisect_right(bases, a) - 1][0] + segs[bisect.bisect_right(bases, a) - 1][1])
regions = []
for j, a in enumerate(addrs):
    sj = bisect.bisect_right(bases, a) - 1
    end = segs[sj][0] + segs[sj][1]
    if j + 1 < len(addrs) and addrs[j + 1] < end:
        end = addrs[j + 1]
    a4 = (a + 3) // 4 * 4
    if end - a4 >= 4:
        regions.append((a, a4, (end - a4) // 4 * 4))
total = sum(r[2] for r in regions)
print(f"{stage}: {tape.n_nodes} nodes on {tape.n_lanes} lanes, {len(addrs)} distinct addresses, {len(regions)} regions, {total / 2**20:.0f} MiB compared per replay", flush=True)
views = [RawMem(a4, nb).tensor() for _, a4, nb in regions]
rng = VF.rng_state(eng.dev)
rng0 = rng.clone()


def serial():
    rng.copy_(rng0)
    for i in range(tape.n_nodes):
        H.call("vx_tape_launch_node", tape.handle, i, H.stream_ptr())
    torch.cuda.synchronize()


serial()
ref = [v.clone() for v in views]
serial()
unstable = set(j for j, (v, r) in enumerate(zip(views, ref)) if not torch.equal(v, r))
print(f"{len(unstable)} regions differ between two serial launches (float atomics, scratch): ignored", flush=True)
if os.environ.get("VX_MEMDIFF_ONLY"):
    # the first K nodes only: concurrently (no tail) against the same K nodes one after the other; every region that differs, with the nodes that reference it
    K = int(os.environ["VX_MEMDIFF_ONLY"])
    rng.copy_(rng0)
    for i in range(K):
        H.call("vx_tape_launch_node", tape.handle, i, H.stream_ptr())
    torch.cuda.synchronize()
    refk = [v.clone() for v in views]
    for t in range(tries):
        rng.copy_(rng0)
        H.call("vx_tape_replay_prefix", tape.handle, H.stream_ptr(), -K)
        torch.cuda.synchronize()
        bad = [j for j, (v, r) in enumerate(zip(views, refk)) if not torch.equal(v, r)]
        if bad:
            print(f"--- try {t}: first {K} nodes concurrently: {len(bad)} regions differ from the same nodes launched one after the other")
            for j in bad:
                a, a4, nb = regions[j]
                d = torch.nonzero(views[j] != refk[j]).flatten()
                vf, rf_ = views[j].view(torch.float32), refk[j].view(torch.float32)
                who = "; ".join(f"n{i} L{lane[i]} {names[i][:36]} p{k}{'c' if const else ''}" for (i, k, ty, const) in sorted(refs[a]) if i < K)
                print(f"  {a:#x} +{nb:>9d} B: {d.numel():>8d} words differ [{int(d[0])} .. {int(d[-1])}] | {who}")
                print(f"      first words {d[:8].tolist()}: got {[f'{v:.7g}' for v in vf[d[:8]].tolist()]} serial {[f'{v:.7g}' for v in rf_[d[:8]].tolist()]}")
                dd = d.tolist()
                runs, start, prev = [], dd[0], dd[0]
                for w_ in dd[1:]:
                    if w_ != prev + 1:
                        runs.append((start, prev)); start = w_
                    prev = w_
                runs.append((start, prev))
                print(f"      {len(runs)} runs of consecutive words; first runs: {runs[:10]}")
            break
    else:
        print(f"first {K} nodes concurrently: no deviation in {tries} tries")
    sys.exit(0)
if os.environ.get("VX_MEMDIFF_PREFIX"):
    # binary search: the smallest k such that running the first k nodes concurrently (the rest serially) deviates from the all-serial state in some of `ntrial` tries
    ntrial = int(os.environ["VX_MEMDIFF_PREFIX"])

    def deviates(k):
        for _ in range(ntrial):
            rng.copy_(rng0)
            H.call("vx_tape_replay_prefix", tape.handle, H.stream_ptr(), int(k))
            torch.cuda.synchronize()
            if any(j not in unstable and not torch.equal(v, r) for j, (v, r) in enumerate(zip(views, ref))):
                return True
        return False
    lo, hi = 0, tape.n_nodes
    print(f"prefix 0: {deviates(0)}, prefix {hi}: {deviates(hi)}", flush=True)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        dv = deviates(mid)
        print(f"  first {mid} nodes concurrent: {'deviates' if dv else 'equal to serial'}", flush=True)
        if dv:
            hi = mid
        else:
            lo = mid
    print(f"the deviation needs node {hi - 1} in the concurrent part:")
    for i in range(max(0, hi - 14), min(tape.n_nodes, hi + 3)):
        print(f"   {'>>' if i == hi - 1 else '  '} node {i} lane {lane[i]} waits {waits[i]} grid {grid[i]} {names[i]}")
    sys.exit(0)
hits = 0
for t in range(tries):
    rng.copy_(rng0)
    tape.replay()
    torch.cuda.synchronize()
    bad = [j for j, (v, r) in enumerate(zip(views, ref)) if j not in unstable and not torch.equal(v, r)]
    if not bad:
        continue
    hits += 1
    print(f"--- replay {t}: {len(bad)} regions differ from the serial launch (sorted by the first node that references them)", flush=True)
    bad.sort(key=lambda j: min(i for i, _k, _ty, _c in refs[regions[j][0]]))
    for n_, j in enumerate(bad):
        a, a4, nb = regions[j]
        ne = views[j] != ref[j]
        d = torch.nonzero(ne).flatten()
        i0, i1 = int(d[0]), int(d[-1])
        vf, rf_ = views[j].view(torch.float32), ref[j].view(torch.float32)
        fin = torch.isfinite(vf) & torch.isfinite(rf_)
        mabs = float(((vf - rf_).abs() * fin).max())
        scale = float((rf_.abs() * fin).max())
        first = min(refs[a])
        who = "; ".join(f"n{i} L{lane[i]} {names[i][:36]} p{k}{'c' if const else ''}" for (i, k, ty, const) in sorted(refs[a])[:4])
        print(f"  {a:#x} +{nb:>9d} B: {d.numel():>8d} words differ [{i0} .. {i1}] max|d| {mabs:.3g} of {scale:.3g} | {who}")
        if n_ < 3:
            gi, ri = views[j][d[:6]].tolist(), ref[j][d[:6]].tolist()
            print(f"      as int32   got {gi} serial {ri}")
            print(f"      as float32 got {[f'{v:.7g}' for v in vf[d[:6]].tolist()]} serial {[f'{v:.7g}' for v in rf_[d[:6]].tolist()]}")
    if hits >= 3:
        break
print(f"{wl} B={B} {stage}: {hits} deviating replays in {t + 1}")
