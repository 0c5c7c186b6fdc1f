#!/usr/bin/env python3
"""Where does the taped training step spend its GPU time?  Every node of every stage tape is timed alone (vx_tape_profile), the stage's
critical path is computed over the tape's own schedule (lane order + cross-lane waits) and compared with the measured stage time, and
the kernels ON the critical path are listed -- those are the ones worth making faster.  argv: [workload]"""
import collections, ctypes, os, re, subprocess, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd import _hip as H
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
if os.environ.get("VX_ALL_DROP"):      # what-if: every dropout probability (proj / conv / attn) -> the cost of ALL Philox work of the step
    pd = float(os.environ["VX_ALL_DROP"])
    cfg = dict(cfg, attn_drop=pd, proj_drop=pd, conv_drop=pd)
if os.environ.get("VX_ATTN_DROP"):
    cfg = dict(cfg, attn_drop=float(os.environ["VX_ATTN_DROP"]))      # what-if: cost of the attention dropout (Philox) in the attention kernels
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False, precision=os.environ.get("VX_PRECISION", "fp32"))
for _ in range(3):
    eng.step(x, lab)
torch.cuda.synchronize()
G = eng.graphs
STRIDE = 160


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    n = n.replace("at::native::", "")
    return n[:64]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, timeout=20).stdout.split("\n")
        return out[:len(names)] if len(out) >= len(names) else names
    except Exception:
        return names


def analyse(tag, tape):
    n = tape.n_nodes
    if n == 0 or tape.handle is None:
        return dict(tag=tag, n=0, total=0.0, cp=0.0, lanes={}, path=[], all=[])
    us = (ctypes.c_float * n)()
    H.call("vx_tape_profile", tape.handle, H.stream_ptr(), 3, ctypes.addressof(us))
    lane, grid, waits = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * (4 * n))()
    names = ctypes.create_string_buffer(n * STRIDE)
    H.call("vx_tape_describe", tape.handle, ctypes.addressof(lane), ctypes.addressof(grid), ctypes.addressof(waits), ctypes.addressof(names), STRIDE)
    nm = demangle([names.raw[i * STRIDE:(i + 1) * STRIDE].split(b"\0")[0].decode() for i in range(n)])
    nm = [short(s_) for s_ in nm]
    fin, via, last = [0.0] * n, [-1] * n, {}
    for i in range(n):
        best, arg = 0.0, -1
        cands = [last.get(lane[i], -1)] + [waits[4 * i + w] for w in range(4)]
        for p in cands:
            if p >= 0 and fin[p] > best:
                best, arg = fin[p], p
        fin[i], via[i] = best + us[i], arg
        last[lane[i]] = i
    end = max(range(n), key=lambda i: fin[i])
    path, i = [], end
    while i >= 0:
        path.append(i)
        i = via[i]
    lanes = collections.Counter()
    for i in range(n):
        lanes[lane[i]] += us[i]
    return dict(tag=tag, n=n, total=sum(us), cp=fin[end], lanes=dict(lanes), path=[(nm[i], us[i], grid[i]) for i in reversed(path)], all=[(nm[i], us[i], grid[i]) for i in range(n)],
                sched=[(i, lane[i], nm[i], us[i], fin[i] - us[i], [waits[4 * i + w] for w in range(4) if waits[4 * i + w] >= 0], i in path) for i in range(n)])


def measure(fn):
    blk = torch.randn(8192, 8192, device="cuda")
    tot = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        blk @ blk; blk @ blk
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / 5 * 1e3


stages = [("enc_fwd", [G["enc_fwd"]], lambda: G["enc_fwd"].replay()), ("dec_fwd", G["dec_fwd"], lambda: eng._fan(G["dec_fwd"])), ("loss", [G["loss"]], lambda: G["loss"].replay()),
          ("dec_bwd", G["dec_bwd"], lambda: eng._fan(G["dec_bwd"])), ("enc_bwd", [G["enc_bwd"]], lambda: G["enc_bwd"].replay())]
on_path, everything = collections.Counter(), collections.Counter()
cnt_path = collections.Counter()
step_cp = step_meas = 0.0
verbose = os.environ.get("VX_VERBOSE", "0") == "1"
for tag, tapes, fn in stages:
    res = [analyse(f"{tag}[{k}]", t) for k, t in enumerate(tapes)]
    meas = measure(fn)
    cp = max(r["cp"] for r in res)
    worst = max(res, key=lambda r: r["cp"])
    step_cp += cp; step_meas += meas
    print(f"{tag:8s}: measured {meas:7.1f} us | critical path {cp:7.1f} us | kernel-time sum {sum(r['total'] for r in res):7.1f} us over {sum(r['n'] for r in res)} nodes | lanes "
          + ", ".join("/".join(f"{v:.0f}" for v in r["lanes"].values()) for r in res))
    for nm, u, g in worst["path"]:
        on_path[nm] += u; cnt_path[nm] += 1
    for r in res:
        for nm, u, g in r["all"]:
            everything[nm] += u
    if verbose:
        for nm, u, g in worst["path"]:
            print(f"      {u:7.1f} us  g{g:<6d} {nm}")
    if os.environ.get("VX_VERBOSE", "0") == "2":
        # the whole schedule of the stage's longest tape: node, lane, earliest start, duration, cross-lane waits, * = on the critical path
        for i, ln, nm, u, t0, ws, onp in worst["sched"]:
            print(f"   {'*' if onp else ' '} #{i:<3d} lane {ln}  start {t0:7.1f}  {u:6.1f} us  {nm}" + (f"   waits {ws}" if ws else ""))
print(f"stages one after the other: measured {step_meas:.0f} us, sum of stage critical paths {step_cp:.0f} us")
if "dec_wg" in G:
    # the decoders' weight gradients run on the fourth lane BESIDE the encoder backward: that phase is as long as the two together need
    res = [analyse(f"dec_wg[{k}]", t) for k, t in enumerate(G["dec_wg"])]
    def wg_only():
        for t in G["dec_wg"]:
            t.replay()
    m_wg = measure(wg_only)
    def both():
        lane = eng._lane_streams(4)[3]
        cur = torch.cuda.current_stream()
        eng._hop(40, cur, lane)
        with torch.cuda.stream(lane):
            wg_only()
        G["enc_bwd"].replay()
        eng._hop(41, lane, cur)
    m_both = measure(both)
    # which of the two ends the phase: finish times of the weight-gradient lane and of the encoder backward, from a common start
    fin = []
    for _ in range(20):
        lane = eng._lane_streams(4)[3]
        cur = torch.cuda.current_stream()
        e0, e_wg, e_enc = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        e0.record(cur)
        eng._hop(40, cur, lane)
        with torch.cuda.stream(lane):
            wg_only()
            e_wg.record(lane)
        G["enc_bwd"].replay()
        e_enc.record(cur)
        eng._hop(41, lane, cur)
        torch.cuda.synchronize()
        fin.append((e0.elapsed_time(e_wg) * 1e3, e0.elapsed_time(e_enc) * 1e3))
    fin.sort()
    print(f"          in that phase the weight-gradient lane finishes after {fin[len(fin) // 2][0]:7.1f} us, the encoder backward after {sorted(f[1] for f in fin)[len(fin) // 2]:7.1f} us (medians of 20)")
    print(f"dec_wg  : measured {m_wg:7.1f} us alone | kernel-time sum {sum(r['total'] for r in res):7.1f} us over {sum(r['n'] for r in res)} nodes | dec_wg || enc_bwd together: {m_both:7.1f} us")
    wgk = collections.Counter()
    for r in res:
        for nm, u, g in r["all"]:
            wgk[nm] += u
    for nm, u in wgk.most_common(14):
        print(f"      {u:7.1f} us  {nm}")
m_step = measure(lambda: eng.step(x, lab))
print(f"step: measured {m_step:.0f} us (input copies + tapes + AdamW)")
print("kernels on the critical paths (time on path / time in the whole step):")
for nm, u in on_path.most_common(int(os.environ.get("VX_CP_TOP", "40"))):
    print(f"  {u:7.1f} us {cnt_path[nm]:3d}x / {everything[nm]:7.1f} us  {nm}")
