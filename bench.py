#!/usr/bin/env python3
"""Headline benchmark: VeloxSeg training patches/s on synthetic 128^3 2-modality volumes (BASELINE.json metric).

One "step" = zero_grad + forward + Dice/CE/SDKT loss + backward + [RCCL gradient all-reduce] + AdamW on one batch
of synthetic patches already resident in HBM.  `python bench.py --gpus N --steps K --warmup W`; for N > 1 launch
with torch.distributed.run (one rank per GPU).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = dist = None      # imported in _imports(): the self-launcher for --gpus N > 1 must not touch HIP before it has started its children


def _imports():
    global torch, dist
    import veloxseg_amd  # noqa: F401  (first: configures the HIP runtime before anything initialises it)
    import torch as _t
    import torch.distributed as _d
    torch, dist = _t, _d

LOSS_CFG = {"deep_Loss_weight": [1, 1, 1, 1], "RC_Loss_weight": 0.5, "Feature_Loss_weight": 2.0}   # config/train_config_bs4.json:52-59

BASE = dict(patch_size=4, base_ch=16, conv_depths=[1, 1, 1, 1], kernel_sizes=[1, 3, 5], min_dim_group=[4, 8, 8, 16],
            conv_expansion_factor=[3, 3, 2, 2], attn_base_ch=16, depths=[1, 1, 1, 1], min_small_window_sizes=[[1, 1, 1]] * 4,
            min_dim_head=[4, 8, 8, 16], ffn_expansion_ratio=[3, 3, 2, 2], num_heads=[1, 2, 2, 4], proj_drop=0.1, conv_drop=0.1, spatial_dim=3)
W128 = [[4] * 3, [8] * 3, [4] * 3, [4] * 3]      # [3,6,3,3] does not tile the 32^3 token grid of a 128^3 patch
W96 = [[3] * 3, [6] * 3, [3] * 3, [3] * 3]
WHECK = [[4, 4, 2], [8, 8, 4], [4, 4, 2], [4, 4, 2]]     # config/models_config_hecktor2022.json "VeloxSeg": anisotropic windows on a 128 x 128 x 64 patch (l = 32 / 256 tokens)
WORKLOADS = {
    # name: (model kwargs, default per-GPU batch)
    # B = 4 per GPU = the reference's effective step batch (batch_size 2 x RandCropByPosNegLabeld num_samples 2) and BASELINE configs[2,3]
    "autopet128": (dict(BASE, input_size=[128] * 3, in_ch=[1, 1], n_classes=2, min_big_window_sizes=W128), 4),
    "autopet96": (dict(BASE, input_size=[96] * 3, in_ch=[1, 1], n_classes=2, min_big_window_sizes=W96), 4),
    "brats128": (dict(BASE, input_size=[128] * 3, in_ch=[4], n_classes=4, min_big_window_sizes=W128), 2),
    "brats96": (dict(BASE, input_size=[96] * 3, in_ch=[4], n_classes=4, min_big_window_sizes=W96), 2),
    "hecktor": (dict(BASE, input_size=[128, 128, 64], in_ch=[1, 1], n_classes=2, min_big_window_sizes=WHECK), 4),
}
# Algorithmic work per PATCH of a full training step (FlopCounterMode fwd+bwd; block-boundary activation traffic, fp32),
# and optimizer traffic per STEP (32 B per parameter): (GFLOP / patch, MB / patch, MB optimizer / step)
STEP_WORK = {"autopet128": (58.50, 569.1, 73.0), "autopet96": (24.11, 240.1, 73.0), "brats128": (71.55, 916.5, 60.0), "brats96": (30.04, 386.6, 60.0),
             # hecktor: FlopCounterMode over the oracle's fwd + loss + bwd at (1, 2, 128, 128, 64) (round 6: 28.83 GFLOP); block-boundary elements 16.96 V per sample
             "hecktor": (28.83, 284.6, 73.3)}
EVAL_GFLOP = {"autopet96": 3.50, "autopet128": 8.75, "brats96": 5.13, "brats128": 12.26, "hecktor": 4.19}      # eval forward, 2 x MAC per patch (hecktor: round 6, same method)
PUBLISHED_EVAL = {"autopet96": 599.06}      # README.md:215 (RTX 3090, autocast, batch <= 16, 10 s + 60 s: speed_test.py:117-134) -- the reference's only GPU number
BF16_DTYPE = ("bf16: 16-bit STORAGE of the full-resolution logits / reconstructions and their gradients and of the block-internal tensors (y_k, o, dn, d_o, g_k) of the 32^3-level JLC blocks; "
              "bf16 MFMA operands in the patch-expand layers and the JLC grouped convolutions of the 32^3 / 16^3 levels; fp32: accumulation, InstanceNorm / LayerNorm statistics, soft-max, "
              "loss sums, master weights, flat gradients, AdamW, every block-boundary tensor and everything at <= 16^3 (VELOXSEG_BF16_STORAGE=0: operands only, fp32 storage)")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3       # fp32 vector / fp32-input MFMA peak
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 2:1-sparsity headline is NOT used)


def synth(cfg, B, device, seed):
    if torch is None:
        _imports()
    g = torch.Generator().manual_seed(seed)
    S = cfg["input_size"]
    x = torch.randn(B, sum(cfg["in_ch"]), *S, generator=g)
    if cfg["n_classes"] == 2:
        lab = (torch.rand(B, 1, *S, generator=g) > 0.97).long()
    else:
        lab = torch.randint(0, cfg["n_classes"], (B, 1, *[s // 8 for s in S]), generator=g)
        lab = lab.repeat_interleave(8, 2).repeat_interleave(8, 3).repeat_interleave(8, 4)
    return x.to(device), lab.to(device)


def cpu_baseline(cfg, B, budget_s=25.0):
    """The CPU oracle (a port of the reference algorithm, pinned to it by tests/golden) on this host's cores: same workload,
    bounded sample (1 warm-up step, then steps until ~budget_s)."""
    from oracle import veloxseg_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipe import fill_state_dict
    ocfg = O.OracleConfig(**{**cfg, "attn_drop": 0.1})
    sd = fill_state_dict(O.state_dict_template(ocfg), seed=7)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    full = dict(sd)
    full.update(params)
    opt = torch.optim.AdamW(list(params.values()), lr=2.5e-4, weight_decay=0.01)
    x, lab = synth(cfg, B, "cpu", 12345)
    cores = min(16, os.cpu_count() or 1)     # measured on the GPU box: 16 threads 1.21, 32 -> 1.01, 64 -> 0.49, 128 -> 0.21 patches/s
    torch.set_num_threads(cores)

    def step():
        opt.zero_grad(set_to_none=True)
        outs = O.forward(x, full, ocfg, training=True)
        loss = O.loss(outs, lab, x, ocfg.M, LOSS_CFG)
        loss.backward()
        opt.step()
        return float(loss)

    step()
    t0, n = time.time(), 0
    while True:
        step()
        n += 1
        if time.time() - t0 > budget_s or n >= 20:
            break
    dt = (time.time() - t0) / n
    out = {"value": round(B / dt, 4), "unit": "patches/s", "cores": cores, "kind": "port",
           "sample": f"{n} full train steps (fwd+loss+bwd+AdamW) of the same workload, batch {B}, fp32, after 1 warm-up step",
           "host_cores": os.cpu_count(), "cpu_model": _cpu_model()}
    # the reference's own CPU protocol (speed_test.py:64-70,102-115): eval forward, ONE thread, batch 1 (README.md:216 quotes 6.67 patches/s at 96^3)
    try:
        torch.set_num_threads(1)
        with torch.no_grad():
            x1 = x[:1]
            O.forward(x1, sd, ocfg, training=False)
            t0, n1 = time.time(), 0
            while n1 < 5 and time.time() - t0 < 12.0:
                O.forward(x1, sd, ocfg, training=False)
                n1 += 1
        out["eval_1thread_bs1"] = {"value": round(n1 / (time.time() - t0), 4), "unit": "patches/s", "sample": f"{n1} eval forwards, 1 thread, batch 1 (speed_test.py protocol)"}
    except Exception as e:      # a reported extra, never a reason to lose the line
        out["eval_1thread_bs1"] = {"error": str(e)[:200]}
    finally:
        torch.set_num_threads(cores)
    return out


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def eager_rocm_baseline(cfg, B, dev, budget_s=8.0):
    """PyTorch-ROCm EAGER training step = the oracle's plain aten formulation run on the MI355X (aten / MIOpen / rocBLAS kernels), the
    denominator of north_star's '>= 5x eager' target.  Bounded sample outside the timed region; the oracle is used as a baseline here exactly
    as in cpu_baseline, never as the product."""
    from oracle import veloxseg_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipe import fill_state_dict
    ocfg = O.OracleConfig(**{**cfg, "attn_drop": 0.1})
    sd = {k: v.to(dev) for k, v in fill_state_dict(O.state_dict_template(ocfg), seed=7).items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    full = dict(sd)
    full.update(params)
    opt = torch.optim.AdamW(list(params.values()), lr=2.5e-4, weight_decay=0.01)
    x, lab = synth(cfg, B, dev, 12345)

    def step():
        opt.zero_grad(set_to_none=True)
        outs = O.forward(x, full, ocfg, training=True)
        loss = O.loss(outs, lab, x, ocfg.M, LOSS_CFG)
        loss.backward()
        opt.step()

    step()
    step()
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    while n < 10 and time.perf_counter() - t0 < budget_s:
        step()
        n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"value": round(B / dt, 3), "unit": "patches/s", "ms_per_step": round(dt * 1e3, 2),
            "sample": f"{n} eager steps (aten ops, fp32, same workload / batch / optimizer) after 2 warm-up steps"}


def _self_launch(n):
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them) and wait; rank 0 prints the
    JSON line on the inherited stdout.  The parent never initialises the GPU (no exec of an initialised process, no torch import here)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = p_.wait() or rc
    return rc


def _tapes(eng):
    G = eng.graphs
    return [G["enc_fwd"], G["loss"], G["enc_bwd"]] + list(G["dec_fwd"]) + list(G["dec_bwd"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS))
    ap.add_argument("--mode", default="train", choices=["train", "eval", "sliding"],
                    help="train (default): the headline training step.  eval: the reference's one published GPU protocol (speed_test.py:117-134: eval forward, batch <= 16, "
                         "T0 s warm-up + T1 s timed, a device synchronise per forward; default workload autopet96) through engine.TapedPredictor.  sliding: BASELINE configs[4], "
                         "sliding-window inference of a synthetic (1, 4, 240, 240, 155) volume, overlap 0.5, sw_batch 2, arg-max + BraTS Dice included (utils/inference_brats.py:190-218)")
    ap.add_argument("--t0", type=float, default=10.0, help="eval / sliding: warm-up seconds (speed_test.py:10)")
    ap.add_argument("--t1", type=float, default=60.0, help="eval / sliding: timed seconds (speed_test.py:12); --steps K > 0 with --mode eval / sliding times exactly K forwards / volumes instead")
    ap.add_argument("--roi", type=int, default=128, choices=[96, 128], help="sliding: window edge (96 = the shipped BraTS config: 48 windows; 128: 18 windows)")
    ap.add_argument("--batch", type=int, default=0, help="patches per GPU (default: workload default)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="bf16 = the opt-in mode of BASELINE configs[1]: bf16 MFMA operands (fp32 accumulate, fp32 storage) in the patch-expand layers; a separate line, never the headline")
    ap.add_argument("--eager", action="store_true", help="launch every step through autograd (host-bound: ~10 ms of enqueue per step) instead of replaying the captured launch tapes")
    ap.add_argument("--hipgraph", action="store_true", help="replay the captured stages with hipGraphLaunch instead of the launch tape (slower on ROCm 7.2: DESIGN.md section 3)")
    ap.add_argument("--no-graph", action="store_true", help="(default) eager launches on forked HIP streams")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--pipeline-tail", action="store_true", help="A/B: TrainEngine(pipeline_tail=True) -- the decoders' weight-gradient lane is joined before the NEXT decoder forward instead of at the end of the step (measured: 793 vs 802 patches/s, the GPU is throughput-bound in aggregate; off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-pass", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL over xGMI, production) or gloo (debug: lets several ranks share one GPU)")
    ap.add_argument("--no-eager-baseline", action="store_true")
    ap.add_argument("--dispersion-steps", type=int, default=300, help="extra steps after the timed region whose per-step times give p10 / p50 / p90 (0 = skip)")
    ap.add_argument("--lane-probe", action="store_true", help="diagnostic: after the timed region, time dispatch-heavy kernels on every pair of tape lanes (lanes that share a dispatch pipe overlap worse)")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = {"train": "autopet128", "eval": "autopet96", "sliding": "brats128" if args.roi == 128 else "brats96"}[args.mode]
    if args.steps is None:
        args.steps = 300 if args.mode == "train" else 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))          # plain `python bench.py --gpus N`: one child process per GPU, started before anything touches HIP
    _imports()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the VeloxSeg hot path has no CPU fallback")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()          # debug mode: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"

    import types
    from veloxseg_amd import _hip as H
    from veloxseg_amd import functional as VF
    from veloxseg_amd.engine import TrainEngine
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils.loss import Loss

    cfg, defB = WORKLOADS[args.workload]
    if args.mode != "train":
        assert world == 1, "--mode eval / sliding run on one GPU (replicas only: independent windows / batches, no exchange step)"
        out = (_eval_mode if args.mode == "eval" else _sliding_mode)(args, cfg, dev)
        print(json.dumps(out), flush=True)
        return
    B = args.batch or defB
    torch.manual_seed(12345)                                   # reference seed (utils/seed.py:6)
    model = VeloxSeg(**cfg).to(dev)
    VF.manual_seed(12345 + rank, dev)
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, dev, num_modal=len(cfg["in_ch"]))
    eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=not args.eager, replay="graph" if args.hipgraph else "tape", overlap=not args.no_overlap, precision="bf16" if args.dtype == "bf16" else "fp32",
                      pipeline_tail=args.pipeline_tail)
    x, lab = synth(cfg, B, dev, 12345 + rank)
    eng.step(x, lab)                                           # capture (+ first step)
    # untimed pre-warm (besides the W warm-up steps): the first process on a cold box was seen 10 % below every later one (clocks, page tables,
    # code pages); ~1.5 s of steps, never part of the timed region
    t_pre = time.perf_counter()
    n_pre = 0
    while (n_pre < 150) if world > 1 else (time.perf_counter() - t_pre < 1.5 and n_pre < 400):      # (a step is a collective with world > 1: every rank the same count)
        eng.step()
        n_pre += 1
        if n_pre % 16 == 0:
            torch.cuda.synchronize()
    for _ in range(max(args.warmup - 1, 0)):
        eng.step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    t_enq = time.perf_counter() - t0                          # the host's share: all K steps are enqueued by now (the GPU is still running them)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss = float(eng.loss)
    assert loss == loss, "loss is NaN"

    # ---- communication diagnostics (world > 1, outside the timed region): the same step WITHOUT its collectives (every rank skips them alike), and the collectives
    # of K more steps bracketed by HIP events on the stream they are issued from.  The first multi-GPU run then says by itself whether the all-reduces are hidden:
    # exposed_ms = step - step_no_comm; per bucket: bytes, enqueue -> complete time, the stream (tape lane / comm stream) it ran on; VELOXSEG_COMM_PLACEMENT repeats
    # the run with the collectives enqueued at the other two places profiles/r03_comm_standin_probe.txt measured on one GPU
    comm = None
    if world > 1:
        k2 = max(10, min(args.steps, 100))
        eng.skip_comm = True
        for _ in range(5):
            eng.step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            eng.step()
        barrier()
        tnc = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(tnc, op=dist.ReduceOp.MAX)
        eng.skip_comm = False
        eng.comm_profile = []
        for _ in range(k2):
            eng.step()
        barrier()
        rep = eng.comm_report()
        eng.comm_profile = None
        ms_nc = float(tnc) / k2 * 1e3
        comm = {"backend": args.backend + (" (RCCL over xGMI)" if args.backend == "nccl" else " (debug: host copies, ranks may share a GPU)"), "rccl_ranks": world,
                "placement": getattr(eng, "comm_placement", None), "overlap": bool(eng.overlap), "payload_bytes_per_step": int(eng.flat.numel) * 4,
                "buckets": rep, "steps_sampled": k2, "step_ms_no_comm": round(ms_nc, 3), "exposed_ms": round(dt / args.steps * 1e3 - ms_nc, 3),
                "note": "exposed_ms = timed step - the same step with every collective skipped (both max over ranks); a bucket's ms_mean = HIP events on its issuing "
                        "stream from 'gradients complete' to 'reduced values visible'.  Budget at 8 GPUs for >= 0.9 weak scaling: exposed_ms <= 0.1 x step (DESIGN.md section 4)"}

    # ---- dispersion (outside the timed region): >= 300 more steps, each bracketed by HIP events on the step's stream; p10 / p50 / p90 of the per-step time
    disp = None
    if world == 1 and args.dispersion_steps > 0:
        nd = int(args.dispersion_steps)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nd + 1)]
        evs[0].record()
        for i in range(nd):
            eng.step()
            evs[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nd))
        chunks = [sum(evs[i].elapsed_time(evs[i + 1]) for i in range(c0, c0 + 50)) / 50 for c0 in range(0, nd - 49, 50)]
        disp = {"steps": nd, "step_ms_p10": round(ts[nd // 10], 3), "step_ms_p50": round(ts[nd // 2], 3), "step_ms_p90": round(ts[(nd * 9) // 10], 3),
                "step_ms_min": round(ts[0], 3), "step_ms_max": round(ts[-1], 3), "step_ms_mean": round(evs[0].elapsed_time(evs[nd]) / nd, 4),
                "patches_per_s_over_these_steps": round(B * nd / (evs[0].elapsed_time(evs[nd]) * 1e-3), 2),      # the headline over >= 300 consecutive steps, whatever --steps the caller timed "mean_ms_per_50_steps": [round(c, 3) for c in chunks],
                "lanes_on_distinct_hw_queues": (H.query("vx_tape_lanes_distinct") if (eng.use_graph and getattr(eng, "replay_mode", "") == "tape") else None)}

    lane_probe = None
    if args.lane_probe and rank == 0 and not args.eager:
        import ctypes
        lanes = []
        for k in range(4):
            h = ctypes.c_void_p()
            H.call("vx_tape_lane_stream", H.stream_ptr(), k, ctypes.addressof(h))
            lanes.append(torch.cuda.ExternalStream(h.value, device=dev))
        bufs = [torch.zeros(1 << 22, device=dev) for _ in range(4)]

        def burst(idx):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            e0.record()
            for i in idx:
                lanes[i].wait_stream(cur)
                with torch.cuda.stream(lanes[i]):
                    for _ in range(100):
                        bufs[i].add_(1.0)                 # 4096 small workgroups per launch: dispatch-heavy
            for i in idx:
                cur.wait_stream(lanes[i])
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3
        for _ in range(2):
            single = [round(burst([i]), 1) for i in range(4)]
            pairs = {f"{i}{j}": round(burst([i, j]), 1) for i in range(4) for j in range(i + 1, 4)}
        lane_probe = {"single_us": single, "pair_us": pairs, "all4_us": round(burst([0, 1, 2, 3]), 1)}

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {"metric": "training patches/s on 128^3 2-mod volumes (fwd+loss+bwd+allreduce+AdamW)" if args.workload == "autopet128"
               else f"training patches/s ({args.workload})",
               "value": round(value, 3), "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": _dtype_string(args),
               "data": "synthetic (randn volumes, rand>0.97 labels, random-init weights, seed 12345)",
               "config": {"workload": f"{args.workload}: VeloxSeg in_ch={cfg['in_ch']} n_classes={cfg['n_classes']} patch {cfg['input_size']} "
                                      f"windows {cfg['min_big_window_sizes']} dropout proj/conv/attn 0.1, full SDKT train step",
                          "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                          "arithmetic": ("fp32 storage, fp32 accumulation everywhere.  The kernels that sit on the matrix pipe form every fp32 product from 16-bit pieces of "
                                         "the operands with fp32-level results (errors against fp64 at the level of the fp32 kernels: tests/test_hip_ops_gpu.py, "
                                         "tests/test_fused_blocks_gpu.py): patch-expand forward / input gradient, JLC grouped convolutions and the attention backward at the "
                                         "64- / 512-token windows from two fp16 pieces of the operand scaled by a power of two (22 mantissa bits, exact rescale); the JLC and "
                                         "patch-expand weight gradients from three bf16 pieces.  Every other kernel computes in fp32 (patch-expand pieces knob: %d)"
                                         % _expand_split()) if args.dtype == "f32" and _expand_split() else "fp32",
                          "hip_graph": bool(eng.use_graph), "lanes_on_distinct_hw_queues": (H.query("vx_tape_lanes_distinct") if (eng.use_graph and getattr(eng, "replay_mode", "") == "tape") else None),
                          "lane_calibration_spin_us": (H.query("vx_tape_spin_us") if (eng.use_graph and getattr(eng, "replay_mode", "") == "tape") else None), "lane_on_caller_queue": (H.query("vx_tape_lane_on_caller_queue") if (eng.use_graph and getattr(eng, "replay_mode", "") == "tape") else None), "launch": (("launch tape per captured stage (csrc/tape.hip): %d kernel nodes on up to %d HIP streams, %d cross-stream dependencies (flag kernels: a store on the producing stream, a poll on the waiting one; events with VELOXSEG_TAPE_FLAGS=0)" % (sum(t.n_kernels for t in _tapes(eng)), max(t.n_lanes for t in _tapes(eng)), sum(t.n_events for t in _tapes(eng)))) if getattr(eng, "replay_mode", "") == "tape" else "hipGraph per stage") if eng.use_graph else "eager; decoder branches, encoder conv chain and per-modality PWA halves on forked HIP streams", "final_loss": round(loss, 5)}}
    if rank == 0 and comm is not None:
        out["comm"] = comm
    if rank == 0 and lane_probe is not None:
        out["lane_probe"] = lane_probe
    if rank == 0 and disp is not None:
        out["dispersion"] = disp
    # ---- per-kernel pass (eager, HIP events on the launch stream) + roofline of the dominant kernel -----------------------
    if rank == 0 and not args.no_kernel_pass:
        from veloxseg_amd import functional as VF
        _pmc_traffic.workload = args.workload
        eng.flat.reattach()
        VF.BRANCH_STREAMS = False          # one stream: HIP-event intervals are then the kernels' own durations, not shared-GPU time
        with eng._settings():                  # (the engine's precision: the eager pass reads the process-wide switches it scopes to its own passes)
            eng._fwd_bwd_single()
            H.profile_begin()
            eng._fwd_bwd_single()
            prof = H.profile_end()
        VF.BRANCH_STREAMS = True
        _kernel_pass(out, prof, model, args.workload, B)
    if rank == 0:
        gf, mb, opt_mb = STEP_WORK[args.workload]
        flops = gf * 1e9 * B * world
        byts = (mb * B + opt_mb) * 1e6 * world
        sec = out["ms_per_step"] * 1e-3
        out["step_roofline"] = {"algorithmic_flops": flops, "algorithmic_bytes": byts,
                                "achieved_tflops": round(flops / sec / 1e12, 2), "frac_fp32": round(flops / sec / 1e12 / (FP32_PEAK_TFLOPS * world), 4),
                                "achieved_gbs": round(byts / sec / 1e9, 1), "frac_hbm": round(byts / sec / 1e9 / (HBM_PEAK_GBS * world), 4),
                                "note": "whole step against the fp32 vector/MFMA peak and the HBM peak of the GPUs used (SURVEY.md 8d work model)"}
    if rank == 0:
        st_ = _step_traffic(args.workload, B, None, args.dtype)
        if st_ is not None:
            st_["algorithmic_bytes"] = byts / world
            st_["ratio"] = round(st_["counter_bytes_per_step"] / (byts / world), 2)
            out["step_traffic"] = st_
    if rank == 0 and world == 1 and not args.no_eager_baseline:
        try:
            torch.cuda.empty_cache()
            out["eager_rocm_baseline"] = eager_rocm_baseline(cfg, B, dev)
            out["eager_rocm_baseline"]["speedup"] = round(out["value"] / out["eager_rocm_baseline"]["value"], 1)
        except Exception as e:
            out["eager_rocm_baseline"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, B)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _timed_loop(fn, t0_s, t1_s, steps, sync_each):
    """speed_test.py:117-134: run `fn` for t0_s seconds untimed, then until the per-call times sum to t1_s (or exactly `steps` calls when steps > 0).
    sync_each: a device synchronise inside every timed call (the reference's protocol); else ONE synchronise around the whole timed run (pipelined launches).
    -> (calls, seconds)"""
    torch.cuda.synchronize()
    t = time.time()
    while time.time() - t < t0_s:
        fn()
        torch.cuda.synchronize()
    if sync_each:
        timing = []
        while (len(timing) < steps) if steps > 0 else (sum(timing) < t1_s):
            a = time.time()
            fn()
            torch.cuda.synchronize()
            timing.append(time.time() - a)
        return len(timing), sum(timing)
    n = steps if steps > 0 else None
    if n is None:                                   # size the run from a short probe so that it lasts about t1_s
        a = time.time()
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        n = max(8, int(t1_s / max((time.time() - a) / 8, 1e-6)))
    torch.cuda.synchronize()
    a = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n, time.time() - a


def _cpu_eval_baseline(cfg, budget_s=15.0):
    """the reference's CPU protocol (speed_test.py:64-70,102-115: eval forward, ONE thread, batch 1; README.md:216 quotes 6.67 patches/s at 96^3) on the oracle"""
    from oracle import veloxseg_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipe import fill_state_dict
    ocfg = O.OracleConfig(**cfg)
    sd = fill_state_dict(O.state_dict_template(ocfg), seed=7)
    x1 = torch.randn(1, sum(cfg["in_ch"]), *cfg["input_size"], generator=torch.Generator().manual_seed(12345))
    prev = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        with torch.no_grad():
            O.forward(x1, sd, ocfg, training=False)
            t0, n1 = time.time(), 0
            while n1 < 8 and time.time() - t0 < budget_s:
                O.forward(x1, sd, ocfg, training=False)
                n1 += 1
        return {"value": round(n1 / (time.time() - t0), 4), "unit": "patches/s", "cores": 1, "kind": "port",
                "sample": f"{n1} eval forwards of the same workload, 1 thread, batch 1, fp32 (speed_test.py:27,68-69,102-115), after 1 warm-up forward",
                "host_cores": os.cpu_count(), "cpu_model": _cpu_model()}
    finally:
        torch.set_num_threads(prev)


def _forward_kernel_pass(out, model, x, workload, B):
    """per-kernel table + rooflines of ONE eager eval forward (HIP events around every C-ABI call, one stream)"""
    from veloxseg_amd import _hip as H
    from veloxseg_amd import functional as VF
    _pmc_traffic.workload = None             # the committed PMC passes are of the training step: no traffic figure is attached to these launches
    VF.BRANCH_STREAMS = False
    try:
        with torch.no_grad():
            model(x)
            H.profile_begin()
            model(x)
            prof = H.profile_end()
    finally:
        VF.BRANCH_STREAMS = True
    _kernel_pass(out, prof, model, workload, B)


def _dtype_string(args):
    return "f32" if args.dtype == "f32" else BF16_DTYPE


def _eval_mode(args, cfg, dev):
    """The reference's published GPU protocol (README.md:215 = 599.06 patches/s on an RTX 3090): eval forward, batch 16, T0 s warm-up + T1 s timed with a device
    synchronise per forward (speed_test.py:117-134, there under torch.amp.autocast -- here --dtype bf16 is the reduced-precision leg, f32 the parity mode)."""
    from veloxseg_amd import functional as VF
    from veloxseg_amd.engine import TapedPredictor
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    B = args.batch or 16                                       # speed_test.py:28,30-50: largest power of two <= 16 that fits (16 fits 288 GB with room)
    torch.manual_seed(12345)
    model = VeloxSeg(**cfg).to(dev).eval()
    VF.set_precision("bf16" if args.dtype == "bf16" else "fp32")
    x = torch.randn(B, sum(cfg["in_ch"]), *cfg["input_size"], device=dev)
    pred = TapedPredictor(model)
    with torch.inference_mode():
        y = pred(x)                                            # capture (+ the replay == eager check of TapedPredictor)
        assert bool(torch.isfinite(y).all())
        n_s, sec_s = _timed_loop(lambda: pred(x), args.t0, args.t1, args.steps, sync_each=True)
        n_p, sec_p = _timed_loop(lambda: pred(x), 0.5, min(args.t1, 10.0), args.steps, sync_each=False)
    value = B * n_s / sec_s
    gf = EVAL_GFLOP.get(args.workload)
    pub = PUBLISHED_EVAL.get(args.workload) if B == 16 else None
    out = {"metric": f"eval forward patches/s ({args.workload}; protocol of the reference's speed_test.py)", "value": round(value, 2), "unit": "patches/s", "n_gpus": 1,
           "steps": n_s, "warmup": f"{args.t0} s", "ms_per_step": round(sec_s / n_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": round(value / pub, 2) if pub else None,
           "vs_baseline_note": ("README.md:215 publishes 599.06 patches/s for this protocol on an RTX 3090 (other hardware; autocast)" if pub else None),
           "dtype": _dtype_string(args), "data": "synthetic (randn volumes, random-init weights, seed 12345)",
           "config": {"workload": f"{args.workload}: VeloxSeg in_ch={cfg['in_ch']} n_classes={cfg['n_classes']} patch {cfg['input_size']} windows {cfg['min_big_window_sizes']}, "
                                  f"eval forward (logits of the main head), batch {B}", "batch_per_gpu": B, "protocol": "speed_test.py:117-134: T0 warm-up, then forwards with a device synchronise "
                                  f"after each until their times sum to T1 (T0 = {args.t0} s, T1 = {args.t1} s)" + (f"; --steps {args.steps}" if args.steps > 0 else ""),
                      "launch": "engine.TapedPredictor: the forward of this batch shape captured once, replayed as a launch tape (csrc/tape.hip)"},
           "pipelined": {"value": round(B * n_p / sec_p, 2), "unit": "patches/s", "forwards": n_p,
                         "note": "the same forwards enqueued back to back, ONE synchronise at the end (what a serving loop does); the headline keeps the reference's per-forward synchronise"}}
    if gf:
        out["step_roofline"] = {"algorithmic_flops": gf * 1e9 * B, "achieved_tflops": round(gf * B / (sec_s / n_s) / 1e3, 2),
                                "frac_fp32": round(gf * B / (sec_s / n_s) / 1e3 / FP32_PEAK_TFLOPS, 4), "note": "eval-forward flops (2 x MAC, SURVEY.md 8d) against the fp32 vector / MFMA peak"}
    if not args.no_kernel_pass:
        _forward_kernel_pass(out, model, x, args.workload, B)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_eval_baseline(cfg)
    return out


def _sliding_mode(args, cfg, dev):
    """BASELINE configs[4]: sliding-window inference on a synthetic full BraTS volume (1, 4, 240, 240, 155), overlap 0.5, sw_batch_size 2, arg-max + BraTS Dice included
    (utils/inference_brats.py:190-218 with train_config batch_size 2) -> volumes/s"""
    from veloxseg_amd import functional as VF
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils import inference_runtime as IR
    from veloxseg_amd.utils.metric.metrics_brats import cal_dice
    roi, overlap = int(args.roi), 0.5
    torch.manual_seed(12345)
    model = VeloxSeg(**cfg).to(dev).eval()
    VF.set_precision("bf16" if args.dtype == "bf16" else "fp32")
    vol = torch.randn(1, 4, 240, 240, 155, device=dev)
    label = torch.randint(0, 4, (1, 1, 240, 240, 155), device=dev)
    starts = IR.window_starts((240, 240, 155), (roi,) * 3, IR.scan_interval((240, 240, 155), (roi,) * 3, overlap))
    res = {}

    def one():
        logits, labels = IR.infer_volume(model, vol, (roi,) * 3, 2, overlap)
        res["dice"] = cal_dice(labels, label)
    with torch.inference_mode():
        one()
        n, sec = _timed_loop(one, min(args.t0, 3.0), min(args.t1, 20.0), args.steps, sync_each=True)
    gf = EVAL_GFLOP.get(args.workload)
    out = {"metric": "sliding-window inference volumes/s (4x240x240x155, overlap 0.5, sw_batch 2; arg-max + BraTS Dice included)", "value": round(n / sec, 3), "unit": "volumes/s",
           "n_gpus": 1, "steps": n, "warmup": f"{min(args.t0, 3.0)} s", "ms_per_step": round(sec / n * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": _dtype_string(args), "data": "synthetic (randn volume, random labels, random-init weights, seed 12345)",
           "config": {"workload": f"brats sliding window: volume (1, 4, 240, 240, 155), roi {roi}^3, overlap {overlap}, sw_batch_size 2 -> {len(starts)} windows; VeloxSeg in_ch=[4] n_classes=4 "
                                  f"windows {cfg['min_big_window_sizes']}", "windows": len(starts), "windows_per_s": round(len(starts) * n / sec, 1),
                      "launch": "utils.inference_runtime.infer_volume: window planner on the host, vx_sw_extract / accumulate / finalize (+ fused uint8 arg-max), the window batch's forward replayed as a launch tape"},
           "dice_vs_random_labels": [round(float(v), 4) for v in res["dice"]]}
    if gf:
        fl = gf * 1e9 * len(starts)
        out["step_roofline"] = {"algorithmic_flops": fl, "achieved_tflops": round(fl / (sec / n) / 1e12, 2), "frac_fp32": round(fl / (sec / n) / 1e12 / FP32_PEAK_TFLOPS, 4),
                                "note": "eval-forward flops of every window (2 x MAC) against the fp32 vector / MFMA peak; blending / arg-max / Dice are HBM passes"}
    if not args.no_kernel_pass:
        xw = torch.randn(2, 4, roi, roi, roi, device=dev)
        _forward_kernel_pass(out, model, xw, args.workload, 2)
    if not args.no_cpu_baseline:
        # the oracle's driver (oracle/sliding_window_oracle.py) on the host cores: ONE window batch of the same roi (bounded sample), scaled to the volume's window count
        from oracle import veloxseg_oracle as O
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from recipe import fill_state_dict
        ocfg = O.OracleConfig(**cfg)
        sd = fill_state_dict(O.state_dict_template(ocfg), seed=7)
        cores = min(16, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        xw = torch.randn(2, 4, roi, roi, roi)
        with torch.no_grad():
            O.forward(xw[:1], sd, ocfg, training=False)
            t0 = time.time()
            O.forward(xw, sd, ocfg, training=False)
            dt = time.time() - t0
        out["cpu_baseline"] = {"value": round(1.0 / (dt * len(starts) / 2), 5), "unit": "volumes/s", "cores": cores, "kind": "port",
                               "sample": f"one window batch (2 x 4 x {roi}^3) through the oracle's forward, scaled to the volume's {len(starts)} windows (blending / arg-max not included)",
                               "host_cores": os.cpu_count(), "cpu_model": _cpu_model()}
    return out


def _kernel_pass(out, prof, model, workload, B):
    """per-kernel table of one eager pass (`prof` = _hip.profile_end(): HIP events around every C-ABI call on its launch stream) -> out["kernel_pass"], the `roofline`
    object of the dominant kernel family, `roofline_pwa` (MFMA utilisation of every attention launch) and `roofline_jlc` (the JLC spatial stage at level 1)"""
    from veloxseg_amd import _hip as H      # noqa: F401
    args = type("A", (), {"workload": workload})()
    rows = sorted(((v[1], v[0], k) for k, v in prof.items()), reverse=True)
    total = sum(r[0] for r in rows)
    top = rows[0]
    out["kernel_pass"] = {"total_ms": round(total, 3), "top": [{"entry": r[2][0], "key": list(r[2][1]), "launches": r[1], "ms": round(r[0], 4)} for r in rows[:8]]}
    # `roofline` = the kernel with the most time per step, AGGREGATED BY ENTRY over every shape it is launched with (the top row of the rocprofv3 summary, which is
    # by kernel name): sum of algorithmic flops / sum of launch time.  avg_launch_ms / achieved / frac are measured live (HIP events on the launch stream, above);
    # `profile` repeats them from the committed rocprofv3 --kernel-trace --stats summary so that frac can be recomputed from profiles/ alone.
    alias = {"vx_pwa_attn_fwd_mb": "vx_pwa_attn_fwd", "vx_pwa_attn_bwd_mb": "vx_pwa_attn_bwd", "vx_pwa_attn_bwd_nofold_mb": "vx_pwa_attn_bwd", "vx_pwa_attn_bwd_nofold": "vx_pwa_attn_bwd",
             # (round 5: the operator code calls the entries that take the pieces mode explicitly -- same kernels, one more trailing integer in the key)
             "vx_jlc_wgrad_tz_ns": "vx_jlc_wgrad_tz", "vx_jlc_tz_fwd_ns": "vx_jlc_tz_fwd", "vx_jlc_tz_bwd_ns": "vx_jlc_tz_bwd", "vx_jlc_tz_prep_ns": "vx_jlc_tz_prep"}
    # (round 6: the *_h entries = the same kernels with the storage type of some tensors as trailing flags -- h16 = 0 in the fp32 mode: (entry, trailing integers to drop))
    h_alias = {"vx_jlc_tz_fwd_h": ("vx_jlc_tz_fwd", 2), "vx_jlc_tz_bwd_h": ("vx_jlc_tz_bwd", 2), "vx_jlc_wgrad_tz_h": ("vx_jlc_wgrad_tz", 2), "vx_mlp_fwd_h": ("vx_mlp_fwd", 1),
               "vx_mlp_bwd_h": ("vx_mlp_bwd", 1), "vx_seg_loss_ds_fwd_h": ("vx_seg_loss_ds_fwd", 1), "vx_seg_loss_ds_bwd_h": ("vx_seg_loss_ds_bwd", 1),
               "vx_jlc_mid_fwd_h": ("vx_jlc_mid_fwd", 1), "vx_jlc_mid_bwd_h": ("vx_jlc_mid_bwd", 1), "vx_jlc_gk_h": ("vx_jlc_gk", 1)}

    def _canon(nm, k_):
        if nm in h_alias:
            return h_alias[nm][0], tuple(k_[:-h_alias[nm][1]])
        if nm.endswith("_ns"):
            return alias.get(nm, nm), (tuple(k_[:-1]) if nm != "vx_jlc_tz_prep_ns" else k_)
        return nm, k_
    rows = [(t_, n_, _canon(nm, k_)) for t_, n_, (nm, k_) in rows]
    fam = {}
    for tot_ms, n, (name, key) in rows:
        base = alias.get(name, name)
        rf = roofline_for(base, key, tot_ms / n, model)
        f = fam.setdefault(base, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0, "known": True, "shapes": [], "last": None})
        f["ms"] += tot_ms
        f["n"] += n
        if rf.get("achieved") is None:
            f["known"] = False
        else:
            f["flops"] += rf["algorithmic_flops"] * n
            f["bytes"] += rf["algorithmic_bytes"] * n
            f["shapes"].append({"args": list(key)[:8], "launches": n, "avg_launch_ms": round(tot_ms / n, 5), "frac": rf["frac"], "algorithmic_flops": rf["algorithmic_flops"]})
            f["last"] = rf
    out["kernel_pass"]["families"] = [{"entry": k_, "launches": f["n"], "ms": round(f["ms"], 4)} for k_, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:8]]
    out["roofline"] = None
    for base, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        if not f["known"] or f["last"] is None:
            continue
        rf = dict(f["last"])
        for drop in ("args", "traffic_source", "pairs", "kernels", "mfma"):
            rf.pop(drop, None)
        sec = f["ms"] * 1e-3
        rf.update({"kernel": base, "device_kernel": DEVICE_KERNEL.get(base, base), "launches_per_step": f["n"], "avg_launch_ms": round(f["ms"] / f["n"], 5),
                   "algorithmic_flops": f["flops"] / f["n"], "algorithmic_bytes": f["bytes"] / f["n"], "share_of_step": round(f["ms"] / total, 4), "shapes": f["shapes"]})
        if rf["bound"] == "mfma":
            ach = f["flops"] / sec / 1e12
        else:
            ach = f["bytes"] / sec / 1e9
        rf["achieved"], rf["frac"] = round(ach, 3), round(ach / rf["peak"], 4)
        if "frac_of_fp32_mfma_peak" in rf:
            rf["frac_of_fp32_mfma_peak"] = round(ach / FP32_PEAK_TFLOPS, 4)
        rf["traffic"], tsrc = _pmc_traffic_family(DEVICE_KERNEL.get(base, base), B)
        if tsrc:
            rf["traffic_source"] = tsrc
        pr = _profile_row(DEVICE_KERNEL.get(base, base), args.workload, B)
        if pr is not None:
            pr["achieved"] = round((f["flops"] if rf["bound"] == "mfma" else f["bytes"]) / f["n"] / (pr["avg_launch_ms"] * 1e-3) / (1e12 if rf["bound"] == "mfma" else 1e9), 3)
            pr["frac"] = round(pr["achieved"] / rf["peak"], 4)
            rf["profile"] = pr
        if base == "vx_jlc_wgrad_tz":
            fc = _wg_full_chip(f["shapes"], rf["peak"])
            if fc is not None:
                rf["full_chip"] = fc
        for sh in rf["shapes"]:
            sh.pop("algorithmic_flops", None)
        out["roofline"] = rf
        break
    # PWA attention (north_star: "MFMA utilisation for PWA against gfx950 peak"): every attention launch of the step, forward and backward
    att = []
    for tot_ms, n, (name, key) in rows:
        if name in ("vx_pwa_attn_fwd", "vx_pwa_attn_bwd", "vx_pwa_attn_fwd_mb", "vx_pwa_attn_bwd_mb", "vx_pwa_attn_bwd_nofold_mb", "vx_pwa_attn_bwd_nofold"):
            base = "vx_pwa_attn_fwd" if "fwd" in name else "vx_pwa_attn_bwd"
            ra = roofline_for(base, key, tot_ms / n, model)
            if ra.get("achieved") is not None:
                att.append({"pass": "forward" if "fwd" in name else "backward", "c_qk": key[2], "c_v": key[3], "pairs": ra.get("pairs"), "avg_launch_ms": ra["avg_launch_ms"],
                            "achieved_tflops": ra["achieved"], "frac_fp32_peak": ra["frac"], **(ra.get("mfma") or {})})
    if att:
        out["roofline_pwa"] = att
    # the JLC spatial stage (the reference's Johnson-Lindenstrauss block, conv_blocks.py:51-58) at level 1: its own roofline object with PMC traffic
    jname = "vx_jlc_tz_fwd" if any(name == "vx_jlc_tz_fwd" for _, _, (name, _k) in rows) else "vx_jlc_conv_fwd"
    jl = [(tot_ms / n, n, key) for tot_ms, n, (name, key) in rows if name == jname]
    if jl:
        ms1, n1, key1 = max(jl, key=lambda t: t[2][3] * t[2][4] * t[2][5])
        rj = roofline_for(jname, key1, ms1, model)
        rj["launches_per_step"] = n1
        if rj.get("traffic") is None and jname == "vx_jlc_tz_fwd":
            # north_star: "rocprof-reported achieved HBM GB/s for JLC" -- the counter bytes of THIS launch shape (group width = first template argument of vx_tz_k) from the
            # committed PMC passes over the launch time measured live
            cg = int(key1[1]) // max(int(key1[2]), 1)
            tr, src = _pmc_traffic_family(r"vx_tz_k<%d, \d+, \d+, \d+, false, " % cg, B)
            if tr is not None:
                rj["traffic"], rj["traffic_source"] = tr, src
        if rj.get("traffic") is not None:
            rj["achieved_gbs"] = round(rj["traffic"] / (ms1 * 1e-3) / 1e9, 1)
            rj["frac_hbm"] = round(rj["achieved_gbs"] / HBM_PEAK_GBS, 4)
            if rj.get("algorithmic_bytes"):
                rj["traffic_over_algorithmic"] = round(rj["traffic"] / rj["algorithmic_bytes"], 3)
        out["roofline_jlc"] = rj


def _pmc_file(workload="autopet128", B=4, dtype="f32"):
    """the newest committed PMC summary (tools/pmc_traffic.py) taken on this workload, batch and dtype; files of rounds 1-5 carry no such fields: autopet128, B = 4, f32"""
    import glob
    import re as _re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_traffic*.json")):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if (d.get("workload", "autopet128"), int(d.get("batch", 4)), d.get("dtype", "f32")) != (workload, int(B), dtype):
            continue
        m = _re.match(r"r(\d+)", os.path.basename(f))
        key = (int(m.group(1)) if m else 0, os.path.basename(f))
        if best is None or key > best[0]:
            best = (key, f)
    return best[1] if best else None


PMC_FILE = _pmc_file() or os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
PMC_NAME = "profiles/" + os.path.basename(PMC_FILE)


def _step_traffic(workload, B, trace_steps=None, dtype="f32"):
    """HBM bytes of ONE training step summed over every kernel of the committed PMC passes (FETCH_SIZE + WRITE_SIZE, separate runs, gfx950 correction:
    tools/pmc_traffic.py) of THIS workload, batch and dtype; null when no such passes are committed"""
    try:
        f = _pmc_file(workload, B, dtype)
        if f is None:
            return None
        d = json.load(open(f))
        # forward/backward passes covered by the trace: the launches of a once-per-pass kernel (the optimiser kernel undercounts: capture / self-check passes run no AdamW)
        steps = trace_steps or d.get("passes_in_trace") or next((v["launches_in_trace"] for k, v in d["kernels"].items() if k.startswith("vx_loss_finalize_k")), None)
        if not steps:
            return None
        tot = sum(float(k["hbm_bytes_per_launch_corrected"]) * k["launches_in_trace"] for k in d["kernels"].values() if "hbm_bytes_per_launch_corrected" in k)
        for k in d["kernels"].values():
            for g in k.get("by_grid", {}).values():
                if "hbm_bytes_per_launch_corrected" not in k:
                    tot += float(g["hbm_bytes_per_launch_corrected"]) * g["launches_in_trace"]
        out = {"counter_bytes_per_step": round(tot / steps), "source": "profiles/" + os.path.basename(f), "passes_in_trace": steps}
        if dtype != "f32":                           # the bf16 line carries the fp32 figure of the same workload beside its own (<= 0.6 x asked)
            f32 = _step_traffic(workload, B, None, "f32")
            if f32 is not None:
                out["f32_counter_bytes_per_step"] = f32["counter_bytes_per_step"]
                out["f32_source"] = f32["source"]
                out["vs_f32"] = round(out["counter_bytes_per_step"] / f32["counter_bytes_per_step"], 3)
        return out
    except Exception:
        return None


def _pmc_traffic_by_grid(kernel, B, grid_threads):
    """HBM bytes per launch of `kernel` at the launch geometry `grid_threads` (total work-items), from the committed PMC passes"""
    try:
        d = json.load(open(PMC_FILE))["kernels"][kernel]
        if B != 4 or _pmc_traffic.workload != "autopet128":
            return None, None
        bg = d.get("by_grid", {})
        if grid_threads in ("max", "min") and bg:            # the kernel runs with two launch geometries per step (level 1 / level 2): largest / smallest grid
            key = (max if grid_threads == "max" else min)(bg, key=int)
            return float(bg[key]["hbm_bytes_per_launch_corrected"]), PMC_NAME + f" (grid of {key} work-items)"
        g = bg.get(str(grid_threads))
        if g is not None:
            return float(g["hbm_bytes_per_launch_corrected"]), PMC_NAME + " (by grid size)"
        if "by_grid" not in d:
            return float(d["hbm_bytes_per_launch_corrected"]), PMC_NAME
    except Exception:
        pass
    return None, None


def _pmc_traffic_by_launches(kernel, B, launches_per_step, trace_steps=5):
    """as _pmc_traffic for a kernel that runs with several shapes per step: the entry of the grid size that was launched launches_per_step times per step"""
    try:
        d = json.load(open(PMC_FILE))["kernels"][kernel]
        if B != 4 or _pmc_traffic.workload != "autopet128":
            return None, None
        hits = [g for g in d.get("by_grid", {}).values() if g["launches_in_trace"] == launches_per_step * trace_steps]
        if len(hits) == 1:
            return float(hits[0]["hbm_bytes_per_launch_corrected"]), PMC_NAME + " (by grid size)"
        if "by_grid" in d:
            return None, None
        note = "" if d["launches_in_trace"] == launches_per_step * trace_steps else f" (mean over all {d['launches_in_trace'] // trace_steps} launches of this kernel per step, more than one shape)"
        return float(d["hbm_bytes_per_launch_corrected"]), PMC_NAME + note
    except Exception:
        return None, None


def _pmc_traffic(kernels, B):
    """HBM bytes per launch of the named kernels from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950
    correction applied: profiles/r02_pmc_traffic.json).  Counters cannot be read from inside bench.py; the figure is valid for the workload
    and batch it was collected on (autopet128, B = 4) and null otherwise."""
    try:
        d = json.load(open(PMC_FILE))["kernels"]
        if B != 4 or _pmc_traffic.workload != "autopet128":
            return None, None
        return float(sum(d[k]["hbm_bytes_per_launch_corrected"] for k in kernels)), PMC_NAME
    except Exception:
        return None, None


_pmc_traffic.workload = None


# C-ABI entry -> regular expression of the device kernel(s) behind it in a rocprofv3 kernel summary
DEVICE_KERNEL = {"vx_jlc_wgrad_tz": r"vx_jlc_wg_k<", "vx_jlc_cl_fwd": r"vx_jlc_cl_fwd_k<", "vx_jlc_cl_bwd": r"vx_jlc_cl_bwd_k<", "vx_jlc_tz_fwd": r"vx_tz_k<\d+, \d+, \d+, \d+, false, ", "vx_jlc_tz_bwd": r"vx_tz_k<\d+, \d+, \d+, \d+, true, ", "vx_jlc_conv_fwd": r"vx_jlc_conv_fwd_k<",
                 "vx_jlc_conv_bwd": r"vx_jlc_conv_bwd_k<", "vx_pwa_attn_bwd": r"vx_pwa_attn_bwd_(both|q|kv)_k<|vx_pwa_attn_bwd1h?_k", "vx_pwa_attn_fwd": r"vx_pwa_attn_(mfma_)?fwd_k<",
                 "vx_expand_fwd_mfma_split": r"vx_expand_fwd_split_k<", "vx_expand_bwd_data_mfma_split": r"vx_expand_bwd_data_split_k<", "vx_expand_wgrad_mfma_split": r"vx_expand_wgrad_split_k<",
                 "vx_mlp_fwd": r"vx_mlp_fwd_k<", "vx_mlp_bwd": r"vx_mlp_bwd_k<", "vx_seg_loss_ds_fwd": r"vx_seg_loss_ds_fwd_k<", "vx_seg_loss_ds_bwd": r"vx_seg_loss_ds_bwd_k<",
                 "vx_conv_mfma_fwd": r"vx_conv_mfma_fwd_k<", "vx_conv_mfma_bwd_data": r"vx_conv_mfma_bwd_data_k<"}


def _wg_full_chip(shapes, peak):
    """The JLC weight-gradient kernel with the WHOLE chip (256 blocks per launch) -- the step runs it on half (128 blocks: the other lanes' kernels use the rest and the step
    is 2 % faster).  Same shapes, same launches per step, timed live with HIP events, one launch after the other on an idle GPU."""
    try:
        import torch
        from veloxseg_amd import _hip as H
        d = torch.device("cuda", torch.cuda.current_device())
        st = H.stream_ptr()
        H.call("vx_jlc_wgrad_tz_set_blocks", 256)
        tot_ms = tot_fl = 0.0
        n_all = 0
        rows = []
        for sh in shapes:
            B, C, G, D, Hh, W = [int(v) for v in sh["args"][:6]]
            x = torch.randn(B, C, D, Hh, W, device=d)
            g = torch.randn(3, B, C, D, Hh, W, device=d)
            dws = [torch.zeros(C, C // G, k, k, k, device=d) for k in (1, 3, 5)]
            n1 = B * C * D * Hh * W
            gp = g.data_ptr()
            run = lambda: H.call("vx_jlc_wgrad_tz_ns", H.P(x), gp, gp + 4 * n1, gp + 8 * n1, H.P(dws[0]), H.P(dws[1]), H.P(dws[2]), B, C, G, D, Hh, W, 22, st)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 40
            rows.append({"args": [B, C, G, D, Hh, W], "avg_launch_ms": round(ms, 5)})
            tot_ms += ms * sh["launches"]
            tot_fl += sh["algorithmic_flops"] * sh["launches"]
            n_all += sh["launches"]
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        return {"blocks_per_launch": 256, "avg_launch_ms": round(tot_ms / n_all, 5), "achieved": round(ach, 3), "frac": round(ach / peak, 4), "shapes": rows,
                "note": "the same kernel given every CU (vx_jlc_wgrad_tz_set_blocks(256)); the step's default is 128 blocks per launch"}
    except Exception as e:          # never let the extra figure take the bench line down
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    finally:
        try:
            H.call("vx_jlc_wgrad_tz_set_blocks", 0)
        except Exception:
            pass


def _latest_profile(pattern):
    import glob
    import re as _re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", pattern)):
        if _re.search(r"autopet96|brats|hecktor", os.path.basename(f)):      # summaries of the other workloads (profiles/r06_kernel_stats_autopet96_*.csv): not the headline's
            continue
        m = _re.match(r"r(\d+)", os.path.basename(f))
        if m and (best is None or (int(m.group(1)), f) > best):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def _profile_row(kernel_re, workload, B):
    """average launch duration of the device kernel(s) matching `kernel_re` in the newest committed rocprofv3 --kernel-trace --stats summary of this bench
    (profiles/rNN_kernel_stats*.csv; taken on the headline workload: autopet128, B = 4) -- lets `roofline.frac` be recomputed from profiles/ alone"""
    import csv
    import re as _re
    try:
        if workload != "autopet128" or B != 4:
            return None
        f = _latest_profile("r*_kernel_stats*.csv")
        if f is None:
            return None
        calls = ns = 0
        names = []
        for row in csv.DictReader(open(f)):
            if _re.search(kernel_re, row["Name"]):
                calls += int(row["Calls"])
                ns += float(row["TotalDurationNs"])
                names.append(_re.sub(r"\(.*", "", row["Name"]).replace("void ", ""))
        if not calls:
            return None
        return {"source": "profiles/" + os.path.basename(f), "kernels": names[:4], "calls": calls, "avg_launch_ms": round(ns / calls * 1e-6, 5),
                "note": "kernels of concurrent lanes share the GPU in the profiled run, so this average sits above the stand-alone HIP-event time measured live"}
    except Exception:
        return None


def _pmc_traffic_family(kernel_re, B):
    """HBM bytes per launch (mean over every shape) of the device kernel(s) matching `kernel_re`, from the committed PMC passes"""
    import re as _re
    try:
        if B != 4 or _pmc_traffic.workload != "autopet128":
            return None, None
        d = json.load(open(PMC_FILE))["kernels"]
        tot = n = 0
        for k_, v in d.items():
            if _re.search(kernel_re, k_ + "("):
                tot += float(v["hbm_bytes_per_launch_corrected"]) * v["launches_in_trace"]
                n += v["launches_in_trace"]
        return (tot / n, PMC_NAME + " (mean over all launches of the kernel)") if n else (None, None)
    except Exception:
        return None, None


def _expand_split():
    try:
        from veloxseg_amd import functional as VF
        m = VF.cpp_module()
        return int(m.get_expand_split()) if m is not None else 0
    except Exception:
        return 0


def _conv_out(d, K, S, P):
    return (d + 2 * P - K) // S + 1


def _attn_mfma_util(name, plan, B, M, cq, cv, pairs, ms):
    """MFMA utilisation of an attention launch against the fp32 MFMA peak: `raw` = the algorithmic GEMM flops of the pass, `padded` = the flops the matrix
    cores actually issue (v_mfma_f32_16x16x4_f32 = 2048 flop; head widths below 16 leave rows of the 16 x 16 tiles idle, and windows are padded to 16 tokens).
    `on_mfma`: whether this pass of this geometry runs its GEMMs on MFMA in the default selection (csrc/pwa_mfma.hip): forward when l % 64 == 0; backward = the
    one-pass kernel where selected, else the fp32-VALU kernels (then both figures are 0 by definition and the VALU rate is the launch's achieved TFLOP/s)."""
    try:
        from veloxseg_amd import _hip as H
        pp = H.ctypes.addressof(plan)
        fwd = name == "vx_pwa_attn_fwd"
        on = bool(H.query("vx_pwa_attn_mfma_ok", pp, B, M, cq, cv) & 1) if fwd else bool(H.query("vx_pwa_attn_bwd1_ok", pp, B, M, cq, cv))
        lpad = (plan.l + 15) // 16 * 16
        tiles = B * plan.heads * plan.Ntot * (M * lpad // 16) ** 2
        if not fwd and H.query("vx_pwa_attn_bwd1h_ok", pp, B, M, cq, cv) == 1:
            # the one-pass backward on the f16 pipe (vx_pwa_attn_bwd1h_k): per 16 x 16 score tile 1 (S) + 1 (dP) + 1 (dV: 2 per key tile and 32 queries) + 1 (dK) +
            # 1.5 (dQ: 3 per query tile and 32 keys) v_mfma_f32_16x16x32_f16 of 16384 flop; the reduction dimension of S / dP is the head width (4 / 8) times the
            # four piece products, the rest of the 32 slots is padding -- priced against the dense 16-bit peak
            raw = pairs * (6.0 * cq + 4.0 * cv)
            padded = tiles * 5.5 * 16384.0
            sec = ms * 1e-3
            return {"on_mfma": True, "pipe": "v_mfma_f32_16x16x32_f16, two scaled fp16 pieces per operand", "mfma_per_tile": 5.5, "tiles": tiles,
                    "mfma_util_raw": round(raw / sec / 1e12 / BF16_PEAK_TFLOPS, 4), "mfma_util_padded": round(padded / sec / 1e12 / BF16_PEAK_TFLOPS, 4),
                    "fp32_equivalent_util": round(raw / sec / 1e12 / FP32_PEAK_TFLOPS, 4),
                    "note": "utilisations against the dense 16-bit peak (2500 TFLOP/s); fp32_equivalent_util = algorithmic flops against the fp32 MFMA peak, "
                            "comparable with the fp32 kernels' figures"}
        cvb = (cv + 15) // 16
        per_tile = (cq // 4 + 4 * cvb) if fwd else (cq // 4 + cv // 4 + 4 * cvb + 4 + 4)      # S (+ dP) k-steps, then 4 k-steps per 16-row output block of PV / dV, dK, dQ
        raw = pairs * ((2.0 * cq + 2.0 * cv) if fwd else (6.0 * cq + 4.0 * cv))
        padded = tiles * per_tile * 2048.0
        sec = ms * 1e-3
        return {"on_mfma": on, "mfma_util_raw": round(raw / sec / 1e12 / FP32_PEAK_TFLOPS, 4) if on else 0.0,
                "mfma_util_padded": round(padded / sec / 1e12 / FP32_PEAK_TFLOPS, 4) if on else 0.0, "mfma_per_tile": per_tile, "tiles": tiles}
    except Exception as e:
        return {"error": str(e)[:120]}


def roofline_for(name, key, ms_per_launch, model=None):
    """Algorithmic flops / bytes of ONE launch of C-ABI entry `name` with integer arguments `key` (argument order: include/veloxseg_hip.h).
    Convolutions and attention sit above the fp32 ridge (157.3 TFLOP/s / 8 TB/s = 20 flop/B) -> priced against the fp32 vector = f32-MFMA peak;
    everything else against HBM."""
    r = {"kernel": name, "args": list(key), "avg_launch_ms": round(ms_per_launch, 5), "traffic": None}
    flops = bytes_ = None
    k = list(key)
    try:
        if name in ("vx_conv3d_fwd", "vx_conv3d_bwd_weight", "vx_conv3d_bwd_weight_tiled", "vx_conv3d_bwd_weight_tiled_ws", "vx_conv3d_bwd_data"):
            if name == "vx_conv3d_bwd_data":
                k = k[:-1]
            B, Cin, Di, Hi, Wi, Cout, K, S, P, G, ps = k[-11:]
            Do, Ho, Wo = (_conv_out(d, K, S, P) for d in (Di, Hi, Wi))
            vin, vout, nw = B * Cin * Di * Hi * Wi, B * Cout * Do * Ho * Wo, Cout * (Cin // G) * K ** 3
            flops, bytes_ = 2.0 * vout * (Cin // G) * K ** 3, 4.0 * (vin + vout + nw)
        elif name == "vx_conv_s1":
            B, Cin, Cout, D, H, W, K, G = k[:8]
            v = B * D * H * W
            flops, bytes_ = 2.0 * v * Cout * (Cin // G) * K ** 3, 4.0 * (v * (Cin + Cout) + Cout * (Cin // G) * K ** 3)
        elif name in ("vx_expand_bwd_data_mfma", "vx_expand_wgrad_mfma"):
            B, Cc, D, H, W = k[:5]
            v = B * D * H * W
            flops, bytes_ = 2.0 * v * 64 * Cc * 16 * 27, 4.0 * (v * (16 + 64 * Cc) + 64 * Cc * 16 * 27)
        elif name in ("vx_expand_fwd_mfma_split", "vx_expand_bwd_data_mfma_split", "vx_expand_wgrad_mfma_split"):
            # the fp32-mode default of the patch-expand layers: every fp32 operand = ns bf16 pieces, a product = 6 (ns = 3) or 3 (ns = 2) bf16 MFMAs with fp32
            # accumulation (csrc/expand_mfma.hip).  Algorithmic flops = the layer's fp32 flops; the ceiling of THIS algorithm on the matrix pipe is the dense
            # bf16 peak divided by the piece products per pair
            ns = int(k[-1])
            B, Cc, D, H, W = k[-7:-2] if name == "vx_expand_bwd_data_mfma_split" else k[-6:-1]
            v = B * D * H * W
            flops, bytes_ = 2.0 * v * 64 * Cc * 16 * 27, 4.0 * (v * (16 + 64 * Cc) + 64 * Cc * 16 * 27)
            if ns == 22 and name == "vx_expand_wgrad_mfma_split":
                ns = 3                                 # (the weight gradient keeps three bf16 pieces under the fp16 mode of the forward / input gradient)
            r["split"] = {"pieces": ns, "bf16_mfma_per_pair": 6 if ns == 3 else 3, "peak_tflops": round(BF16_PEAK_TFLOPS / (6 if ns == 3 else 3), 1)}
            targ = "2, true" if ns == 22 else (f"{ns}, false" if name != "vx_expand_wgrad_mfma_split" else f"{ns}")      # (kernel template arguments: pieces[, fp16 mode])
            kn = {"vx_expand_fwd_mfma_split": f"vx_expand_fwd_split_k<{targ}>", "vx_expand_bwd_data_mfma_split": f"vx_expand_bwd_data_split_k<{targ}>",
                  "vx_expand_wgrad_mfma_split": f"vx_expand_wgrad_split_k<{targ}>"}[name]
            r["traffic"], src = _pmc_traffic_by_grid(kn, B, "max" if Cc >= 2 else "min")
            if src:
                r["traffic_source"] = src
        elif name in ("vx_pwa_attn_fwd", "vx_pwa_attn_bwd"):
            B, M, cq, cv = k[:4]           # (B, M, cq, cv[, dropout site]); windows / tokens come from the model's PWA plan with these head widths
            plan = next((m.plan for m in model.modules() if hasattr(m, "plan") and getattr(m, "c_qk", None) == cq and getattr(m, "c_v", None) == cv), None)
            if plan is not None:
                rows = B * plan.heads * plan.Ntot * M * plan.l
                pairs = rows * M * plan.l
                # forward: QK^T + PV.  backward: S (recomputed from the saved LSE), dP, dV, dQ, dK, each counted ONCE = 6 c_qk + 4 c_v flop per pair (80 at
                # head widths 8 / 8); the VALU kernels compute S and dP twice (once per pass) -- that redundancy is not algorithmic work
                flops = pairs * (2.0 * cq + 2.0 * cv) if name == "vx_pwa_attn_fwd" else pairs * (6.0 * cq + 4.0 * cv)
                bytes_ = 4.0 * rows * ((2 * cq + 2 * cv + 1) if name == "vx_pwa_attn_fwd" else (4 * cq + 4 * cv + 3))
                r["pairs"], r["kernels"] = pairs, 1 if name == "vx_pwa_attn_fwd" else 2
                r["mfma"] = _attn_mfma_util(name, plan, B, M, cq, cv, pairs, ms_per_launch)
                # (the kernels behind the entry: MFMA forward where the geometry allows it; the one-launch backward, else its two passes)
                cands = ([[f"vx_pwa_attn_mfma_fwd_k<{cq}, {cv}>"], [f"vx_pwa_attn_fwd_k<{cq}, {cv}>"]] if name == "vx_pwa_attn_fwd" else
                         [[f"vx_pwa_attn_bwd1h_k<{cq}, {cv}, true>"], [f"vx_pwa_attn_bwd1h_k<{cq}, {cv}, false>"], [f"vx_pwa_attn_bwd_both_k<{cq}, {cv}>"],
                          [f"vx_pwa_attn_bwd_q_k<{cq}, {cv}>", f"vx_pwa_attn_bwd_kv_k<{cq}, {cv}>"]])
                for kn in cands:
                    r["traffic"], r["traffic_source"] = _pmc_traffic(kn, B)
                    if r["traffic"] is not None:
                        break
        elif name in ("vx_jlc_conv_fwd", "vx_jlc_conv_bwd"):
            B, C, G, D, H, W = k[:6]                  # grouped k = 1, 3, 5 convolutions of one JLC block from one K = 5 halo (csrc/jlc.hip)
            v = B * D * H * W
            flops = 2.0 * v * C * (C // G) * (1 + 27 + 125)
            # forward: x read, y1 / y3 / y5 written; backward: g1 / g3 / g5 and d_o read, dx written (+ the three weight tensors)
            bytes_ = 4.0 * (v * C * (4 if name == "vx_jlc_conv_fwd" else 5) + C * (C // G) * 153)
            r["traffic"], src = _pmc_traffic_by_grid("vx_jlc_conv_fwd_k<4>" if name == "vx_jlc_conv_fwd" else "vx_jlc_conv_bwd_k<4>", B, "max" if D * H * W >= 16384 else "min")
            if src:
                r["traffic_source"] = src
        elif name in ("vx_jlc_tz_fwd", "vx_jlc_tz_bwd", "vx_jlc_wgrad_tz", "vx_jlc_cl_fwd", "vx_jlc_cl_bwd"):
            # the same three grouped convolutions (forward / input gradient / the three weight gradients in one launch) as Toeplitz GEMMs on the bf16 matrix pipe
            # with fp32-exact products (csrc/jlc_mfma.hip): algorithmic flops = the layer's fp32 flops, ceiling = dense bf16 peak / 6 piece products
            B, C, G, D, H, W = k[:6]
            v = B * D * H * W
            flops = 2.0 * v * C * (C // G) * (1 + 27 + 125)
            # forward: x read, y1 / y3 / y5 written; input gradient: g1 / g3 / g5 and d_o read, dx written; weight gradients: x and g1 / g3 / g5 read
            bytes_ = 4.0 * (v * C * (5 if name in ("vx_jlc_tz_bwd", "vx_jlc_cl_bwd") else 4) + C * (C // G) * 153)
            ns = 3
            try:
                from veloxseg_amd import _hip as _H
                ns = int(_H.query("vx_jlc_tz_pieces"))
            except Exception:
                pass
            if name.startswith("vx_jlc_cl_"):
                ns = 22                                # csrc/jlc_cl.hip: always two scaled fp16 pieces
            elif name == "vx_jlc_wgrad_tz" and ns == 22:
                ns = 3                                 # (under the fp16 mode of the convolutions the weight gradients keep three bf16 pieces: DESIGN.md section 9)
            npair = {3: 6, 2: 3, 1: 1, 22: 3}[ns]
            r["split"] = {"pieces": ns, "mfma_per_pair": npair, "peak_tflops": round(BF16_PEAK_TFLOPS / npair, 1),
                          "mfma_fill": ("channels-last implicit GEMM: every multiplier of an issue carries a product at group width 16, half of the rows at group width 8"
                                        if name.startswith("vx_jlc_cl_") else
                                        "62.5 % of the multipliers of a k = 5 issue carry a product (37.5 % at k = 3): Toeplitz band 5 of 8")}
        elif name in ("vx_mlp_fwd", "vx_mlp_bwd"):
            ints = [int(a) for a in k]
            B, C, R, V = ints[2:6] if name == "vx_mlp_fwd" else ints[1:5]      # (norm, nparts, B, C, R, V, ...) / (norm, B, C, R, V, ...)
            flops = (4.0 if name == "vx_mlp_fwd" else 10.0) * B * V * C * R      # fwd: two GEMMs; bwd: GEMM1 recomputed + four gradient GEMMs
            bytes_ = 4.0 * (B * V * C * (2 if name == "vx_mlp_fwd" else 3) + 2 * C * R)
        elif name in ("vx_seg_loss_ds_fwd", "vx_seg_loss_ds_bwd"):
            ints = [int(a) for a in k]
            B, C, D, H, W = ints[-5:]
            lab_bytes = {0: 8, 1: 4, 2: 1}.get(ints[1], 8)
            v = B * D * H * W
            low = sum(v // f ** 3 for f in (2, 4, 8)) * C                      # the deep-supervision heads on their own grids
            flops = 40.0 * v * C
            bytes_ = 4.0 * (v * C + low) * (1 if name == "vx_seg_loss_ds_fwd" else 2) + lab_bytes * v
        elif name == "vx_expand_fwd_mfma":
            B, Cc, D, H, W = k[:5]
            v = B * D * H * W
            flops, bytes_ = 2.0 * v * 64 * Cc * 16 * 27, 4.0 * (v * (16 + 64 * Cc) + 64 * Cc * 16 * 27)
        elif name in ("vx_conv_mfma_fwd", "vx_conv_mfma_bwd_data"):
            B, Cin, Di, Hi, Wi, Cout, K, S, P = k[:9]
            Do, Ho, Wo = (_conv_out(d, K, S, P) for d in (Di, Hi, Wi))
            vin, vout, nw = B * Cin * Di * Hi * Wi, B * Cout * Do * Ho * Wo, Cout * Cin * K ** 3
            flops, bytes_ = 2.0 * vout * Cin * K ** 3, 4.0 * (vin + vout + 2 * nw)
        elif name in ("vx_pw_conv_fwd", "vx_pw_conv_bwd_data", "vx_pw_conv_bwd_weight"):
            B, Cin, Cout, V = k[-4:] if name != "vx_pw_conv_bwd_data" else k[-5:-1]
            flops, bytes_ = 2.0 * B * V * Cin * Cout, 4.0 * (B * V * (Cin + Cout) + Cin * Cout)
    except Exception:       # unknown key layout: report the time only
        flops = bytes_ = None
    if flops is None or bytes_ is None:
        r.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None})
        return r
    r["algorithmic_flops"], r["algorithmic_bytes"] = flops, bytes_
    if r.get("split"):
        ach = flops / (ms_per_launch * 1e-3) / 1e12
        pk = r["split"]["peak_tflops"]
        r.update({"bound": "mfma", "achieved": round(ach, 3), "peak": pk, "unit": "TFLOP/s", "frac": round(ach / pk, 4), "frac_of_fp32_mfma_peak": round(ach / FP32_PEAK_TFLOPS, 4),
                  "note": "fp32-accurate products from 16-bit pieces (bf16 x 3 / x 2, or two scaled fp16 pieces): peak = dense 16-bit MFMA peak (%.0f TFLOP/s, MI355X_MICROARCH.md) / %d piece products per pair; "
                          "achieved = the layer's fp32 flops / time" % (BF16_PEAK_TFLOPS, r["split"].get("bf16_mfma_per_pair", r["split"].get("mfma_per_pair")))})
    elif flops / bytes_ > FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        ach = flops / (ms_per_launch * 1e-3) / 1e12
        r.update({"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4),
                  "note": "fp32 kernel above the ridge (%.0f flop/B): fp32 vector = f32-input MFMA peak, MI355X_MICROARCH.md" % (flops / bytes_)})
    else:
        ach = bytes_ / (ms_per_launch * 1e-3) / 1e9
        r.update({"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)})
    return r


if __name__ == "__main__":
    main()
