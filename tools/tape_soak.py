#!/usr/bin/env python3
"""Soak test of the taped step: N replays from the same weights and dropout streams must all give the same loss and gradient (float-atomic noise aside).
A cross-lane dependency that is missed even once shows up as an outlier.  argv: [workload] [batch] [replays]"""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd import functional as VF
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
# A/B: VX_SOAK_OFF=f16bwd,cl,v4 switches the named kernel families back to their predecessors (to find the source of run-to-run deviations); VELOXSEG_STEM_F16=1 turns the
# opt-in stem kernel on
from veloxseg_amd import _hip as H
H.LIB.load()
for name in filter(None, os.environ.get("VX_SOAK_OFF", "").split(",")):
    H.call({"f16bwd": "vx_pwa_attn_set_f16_bwd", "cl": "vx_jlc_cl_set_enabled", "v4": "vx_pw_conv_set_v4"}[name], 0)
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False)
eng.step(x, lab)
torch.cuda.synchronize()
assert eng.use_graph and eng.graphs is not None, "capture failed"
rng = VF.rng_state(eng.dev)
rng0 = rng.clone()
# VX_SOAK_SERIAL=N: first the deterministic audit (veloxseg_amd/tape_audit.py) -- the step node by node on ONE stream in N random orders that respect the tape's
# dependencies; an order whose result deviates is bisected to the unordered pair of kernels
if int(os.environ.get("VX_SOAK_SERIAL", "0")) > 0:
    from veloxseg_amd.tape_audit import serial_audit
    finds, noise = serial_audit(eng, range(1, int(os.environ["VX_SOAK_SERIAL"]) + 1), verbose=bool(os.environ.get("VX_SOAK_VERBOSE")),
                                exhaustive=os.environ.get("VX_SOAK_EXHAUSTIVE") == "1")
    print(f"{wl} B={B}: serial audit, {os.environ['VX_SOAK_SERIAL']} random admissible orders on one stream: {len(finds)} deviate (same-order noise {noise:.1e})")
    for f in finds:
        print("   ", f)
    if finds:
        sys.exit(1)
# VX_SOAK_FUZZ="max_us,prob": every replay puts random spin kernels in front of its nodes (csrc/tape.hip vx_tape_set_fuzz), a different pattern per replay
fuzz = [float(v) for v in os.environ.get("VX_SOAK_FUZZ", "0,0").split(",")]
# VX_SOAK_HYBRID=enc_bwd,dec_wg_beside: only the named stages run concurrently (multi-lane tapes / fans), the others node by node on one stream (tape_audit.hybrid_replay)
hybrid = None
if os.environ.get("VX_SOAK_HYBRID") is not None:
    from veloxseg_amd.tape_audit import hybrid_replay
    hybrid = [k for k in os.environ["VX_SOAK_HYBRID"].split(",") if k]
# Replays alternate between TWO input sets (the volume / labels and their flipped copies, two dropout seeds) unless VX_SOAK_ALTERNATE=0: with identical inputs every
# buffer already holds, from the replay before, exactly what its producer is about to write -- a consumer that runs too early would read the right numbers
alternate = os.environ.get("VX_SOAK_ALTERNATE", "1") != "0"
xA, labA = eng.x.clone(), eng.labels.clone()
xB, labB = xA.flip(2).contiguous(), labA.flip(2).contiguous()
rngB = rng0.clone()
rngB[0] += 12345
refs = [None, None]
ref = None
worst = 0.0
worst_loss = 0.0
bad = 0
ndev = 0
sync_every = int(os.environ.get("VX_SYNC_EVERY", "1"))
for i in range(n):
    which = (i & 1) if alternate else 0
    if alternate:
        eng.x.copy_(xB if which else xA)
        eng.labels.copy_(labB if which else labA)
    rng.copy_(rngB if which else rng0)
    ref = refs[which]
    if fuzz[0] > 0:
        H.call("vx_tape_set_fuzz", i + 1, fuzz[0], fuzz[1])
    if hybrid is not None:
        hybrid_replay(eng, hybrid)
    else:
        eng._replay(comm=False)
    if (i + 1) % sync_every == 0 or i == n - 1:
        torch.cuda.synchronize()
        g = eng.flat.grad
        if ref is None:
            ref = refs[which] = (float(eng.loss), g.clone(), float(g.abs().max()))
        else:
            d = float((g - ref[1]).abs().max()) / ref[2]
            worst = max(worst, d)
            if (os.environ.get("VX_SOAK_VERBOSE") and i < 12) or d > 5e-6:
                j = int((g - ref[1]).abs().argmax())
                name = next((n_ for n_ in eng.flat.names if eng.flat.slices[n_][0] <= j < eng.flat.slices[n_][0] + eng.flat.slices[n_][1]), "?")
                o_, k_ = eng.flat.slices[name] if name != "?" else (0, 0)
                print(f"replay {i}: deviation {d:.3e} at flat index {j} ({name}: element {j - o_} of {k_}; got {float(g[j]):.9e} ref {float(ref[1][j]):.9e})")
                if d > 5e-6:
                    ndev += 1
                if d > 5e-6 and not os.environ.get("VX_SOAK_QUIET"):
                    idx = torch.nonzero((g - ref[1]).abs() > 2e-6 * ref[2]).flatten().tolist()
                    print(f"    {len(idx)} elements deviate by more than 2e-6 of the largest gradient: {idx[:24]}")
                    dd = (g - ref[1]).abs() / ref[2]
                    per = sorted(((float(dd[eng.flat.slices[n_][0]:eng.flat.slices[n_][0] + eng.flat.slices[n_][1]].max()), n_) for n_ in eng.flat.names), reverse=True)
                    print("    largest deviation per parameter: " + "; ".join(f"{n_} {v:.1e}" for v, n_ in per[:14]))
                    clean = [n_ for v, n_ in per if v < 1e-6 and ("layers.2" in n_ or "layers.3" in n_ or "layer3" in n_ or "layer4" in n_)]
                    print(f"    level-3 / level-4 encoder parameters WITHOUT a deviation: {len(clean)}: {clean[:40]}")
                    if os.environ.get("VX_SOAK_STOP"):
                        break
            dl = abs(float(eng.loss) - ref[0]) / max(abs(ref[0]), 1e-30)          # (the loss sums are float atomics too: last-bit differences are noise, not outliers)
            worst_loss = max(worst_loss, dl)
            if d > 1e-4 or dl > 2e-6:
                bad += 1
                print(f"replay {i}: loss {float(eng.loss)} vs {ref[0]}, max |dgrad| / max|grad| = {d:.3e}")
H.call("vx_tape_set_fuzz", 0, 0.0, 0.0)
print(f"{wl} B={B}: {'hybrid ' + ','.join(hybrid) + ': ' if hybrid is not None else ''}{n} replays{' (fuzzed: up to %g us in front of %g of the nodes)' % (fuzz[0], fuzz[1]) if fuzz[0] > 0 else ''}, {bad} outliers, {ndev} replays deviate by more than 5e-6, worst relative gradient deviation {worst:.3e}, worst relative loss deviation {worst_loss:.1e}")
sys.exit(1 if (bad or ndev) else 0)
