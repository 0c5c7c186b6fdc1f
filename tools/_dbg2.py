import os, sys, types, ctypes, struct
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd import _hip as H, functional as VF
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
from veloxseg_amd.tape_audit import *
cfg, B = WORKLOADS["autopet128"]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False)
eng.step(x, lab)
torch.cuda.synchronize()
tape = eng.graphs["enc_fwd"]
lane, waits, names, grid = tape_layout(tape)
n9 = next(i for i in range(tape.n_nodes) if "vx_ln_pw_fwd_k" in names[i])
print("node", n9, names[n9])
buf = (ctypes.c_ubyte * 4096)(); got = ctypes.c_int()
H.call("vx_tape_node_param", tape.handle, n9, 0, ctypes.addressof(buf), 4096, ctypes.addressof(got))
raw = bytes(buf[:got.value])
mods = []
for m in range(2):
    mods.append(struct.unpack_from("<13Q", raw, 104 * m))
C, NS, J0, J1, J2 = struct.unpack_from("<5i", raw, 416)
V, = struct.unpack_from("<q", raw, 440)
tiles_per_b, eps, s2d, C0, gh, gw = struct.unpack_from("<ifiiii", raw, 448)
print("C", C, "NS", NS, "J", J0, J1, J2, "V", V, "tiles_per_b", tiles_per_b, "eps", eps, "s2d", s2d)
def T(ptr, *shape, dtype=torch.float32):
    n = 1
    for s in shape: n *= s
    return RawMem(ptr, 4 * n).tensor().view(dtype).view(*shape)
K = n9 + 2
rng = VF.rng_state(eng.dev); rng0 = rng.clone()
def views():
    out = {}
    for m in range(2):
        xp, gp, bp, w0, w1, w2, b0, b1, b2, xn, o0, o1, o2 = mods[m]
        out[m] = dict(x=T(xp, B, C, V), gamma=T(gp, C), beta=T(bp, C), w=[T(w0, J0, C), T(w1, J1, C), T(w2, J2, C)], b=[T(b0, J0), T(b1, J1), T(b2, J2)],
                      xn=T(xn, B, C, V) if xn else None, out=[T(o0, B, J0, V), T(o1, B, J1, V), T(o2, B, J2, V)])
    return out
vw = views()
def truth(m):
    d = vw[m]
    xx = d["x"].double()
    mu = xx.mean(1, keepdim=True); var = ((xx - mu) ** 2).mean(1, keepdim=True)
    xn = (xx - mu) / torch.sqrt(var + eps) * d["gamma"].double().view(1, C, 1) + d["beta"].double().view(1, C, 1)
    outs = [torch.einsum("jc,bcv->bjv", d["w"][s].double(), xn) + d["b"][s].double().view(1, -1, 1) for s in range(3)]
    return xn, outs
def snap():
    return {m: dict(x=vw[m]["x"].clone(), xn=vw[m]["xn"].clone() if vw[m]["xn"] is not None else None, out=[o.clone() for o in vw[m]["out"]]) for m in range(2)}
rng.copy_(rng0)
for i in range(K):
    H.call("vx_tape_launch_node", tape.handle, i, H.stream_ptr())
torch.cuda.synchronize()
S = snap()
tr = {m: truth(m) for m in range(2)}
for m in range(2):
    e_xn = float((S[m]["xn"].double() - tr[m][0]).abs().max())
    e_o = [float((S[m]["out"][s].double() - tr[m][1][s]).abs().max()) for s in range(3)]
    print("serial m=%d: xn err %.3e out errs %s" % (m, e_xn, ["%.3e" % v for v in e_o]))
for t in range(int(os.environ.get("VX_TRIES", "12"))):
    rng.copy_(rng0)
    H.call("vx_tape_replay_prefix", tape.handle, H.stream_ptr(), -K)
    torch.cuda.synchronize()
    Cn = snap()
    rep = []
    for m in range(2):
        if not torch.equal(Cn[m]["x"], S[m]["x"]): rep.append(f"m{m} INPUT x differs")
        if not torch.equal(Cn[m]["xn"], S[m]["xn"]): rep.append("m%d xn differs: %d words" % (m, int((Cn[m]["xn"] != S[m]["xn"]).sum())))
        for s in range(3):
            ne = Cn[m]["out"][s] != S[m]["out"][s]
            if bool(ne.any()):
                idx = torch.nonzero(ne)
                dv = (Cn[m]["out"][s] - S[m]["out"][s])[ne]
                e_c = float((Cn[m]["out"][s].double() - tr[m][1][s]).abs().max()); e_s = float((S[m]["out"][s].double() - tr[m][1][s]).abs().max())
                rep.append(f"m{m} out{s}: {int(ne.sum())} words, diff min {float(dv.min()):.3e} max {float(dv.max()):.3e}; err vs fp64: concurrent {e_c:.3e} serial {e_s:.3e}; "
                           f"b {sorted(set(idx[:,0].tolist()))} j {sorted(set(idx[:,1].tolist()))} v%4 {sorted(set((idx[:,2] % 4).tolist()))} v range {int(idx[:,2].min())}..{int(idx[:,2].max())} tiles {sorted(set((idx[:,2] // 64).tolist()))[:12]}")
    if rep:
        print(f"try {t}:"); [print("   ", r) for r in rep]
