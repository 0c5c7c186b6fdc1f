#!/usr/bin/env python3
"""Per-stage GPU time of the taped training step (TrainEngine(use_graph=True, replay="tape")): every stage alone, the decoder fans with
their three tapes concurrently (as the step runs them) and one after the other, and the host time of one whole replay."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd import _hip as H
from veloxseg_amd import functional as VF
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False, tape_lanes=int(os.environ.get("VX_LANES", "6")))
for _ in range(3):
    eng.step(x, lab)
torch.cuda.synchronize()
G = eng.graphs


BLOCK = torch.randn(8192, 8192, device="cuda")


def timed(fn, n=20):
    if os.environ.get("VX_PREQUEUE", "1") == "1":      # the whole stage is enqueued while a ~10 ms matmul holds the GPU: the host's ~2.6 us per launch paces nothing
        tot = 0.0
        t_host = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                BLOCK @ BLOCK
            e0.record()
            t0 = time.perf_counter()
            fn()
            t_host += time.perf_counter() - t0
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / 5, t_host / 5 * 1e3
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n * 1e3
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, host


def seq(tapes):
    for t in tapes:
        t.replay()


# NOTE: stages replayed out of order compute garbage but launch the same kernels on the same buffers -- only the clock is read here
rows = [("enc_fwd", lambda: G["enc_fwd"].replay()), ("dec_fwd fan", lambda: eng._fan(G["dec_fwd"])), ("dec_fwd one by one", lambda: seq(G["dec_fwd"])),
        ("loss", lambda: G["loss"].replay()), ("dec_bwd fan", lambda: eng._fan(G["dec_bwd"])), ("dec_bwd one by one", lambda: seq(G["dec_bwd"])),
        ("enc_bwd", lambda: G["enc_bwd"].replay()), ("whole replay", lambda: eng._replay(comm=False))]
for k in range(len(G["dec_fwd"])):
    rows.append((f"dec_fwd[{k}]", lambda k=k: G["dec_fwd"][k].replay()))
    rows.append((f"dec_bwd[{k}]", lambda k=k: G["dec_bwd"][k].replay()))
for name, fn in rows:
    gpu, host = timed(fn)
    print(f"{name:22s} gpu {gpu:7.3f} ms   host {host:6.3f} ms")
print("lanes measured distinct:", H.query("vx_tape_lanes_distinct"))
for name in ("enc_fwd", "loss", "enc_bwd"):
    t = G[name]
    print(name, "nodes", t.n_nodes, "kernels", t.n_kernels, "lanes", t.n_lanes, "events", t.n_events)
for k, t in enumerate(G["dec_fwd"] + G["dec_bwd"]):
    print("dec", k, "nodes", t.n_nodes, "lanes", t.n_lanes, "events", t.n_events)

if os.environ.get("VX_PAIRS", "0") == "1":
    # which streams overlap?  dec_fwd[1] and dec_fwd[2] (independent, ~0.5 ms each) on every pair of 8 fresh streams
    ss = [torch.cuda.Stream() for _ in range(8)]
    cur = torch.cuda.current_stream()

    def pair(a, b):
        def fn():
            for s_, t in ((a, G["dec_fwd"][1]), (b, G["dec_fwd"][2])):
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    t.replay()
            cur.wait_stream(a)
            cur.wait_stream(b)
        return fn
    print("pair overlap (ms; ~1.04 = serial):")
    for i in range(8):
        print("  ", " ".join(f"{timed(pair(ss[i], ss[j]))[0]:5.2f}" if j > i else "  -  " for j in range(8)))

if os.environ.get("VX_LANEPAIRS", "0") == "1":
    ls = eng._lane_streams(4)
    cur = torch.cuda.current_stream()
    print("lane stream handles", [hex(s.cuda_stream) for s in ls])

    def pair(a, b):
        def fn():
            for s_, t in ((a, G["dec_fwd"][1]), (b, G["dec_fwd"][2])):
                if s_ is not cur:
                    s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    t.replay()
            for s_ in (a, b):
                if s_ is not cur:
                    cur.wait_stream(s_)
        return fn
    for i in range(4):
        print("  ", " ".join(f"{timed(pair(ls[i], ls[j]))[0]:5.2f}" if j > i else "  -  " for j in range(4)))

    def tri(order):
        def fn():
            for s_, t in zip([ls[i] for i in order], G["dec_fwd"]):
                if s_ is not cur:
                    s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    t.replay()
            for s_ in ls:
                if s_ is not cur:
                    cur.wait_stream(s_)
        return fn
    for order in ((0, 1, 2), (1, 2, 3), (3, 2, 1)):
        print("three decoders on lanes", order, f"{timed(tri(order))[0]:.3f} ms")
