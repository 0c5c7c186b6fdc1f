"""Data-parallel engine on real kernels: two processes (world_size 2, gloo backend so that both can share the one MI355X of the
test box; the production backend is "nccl" = RCCL) run TrainEngine -- eager launches and per-stage hipGraphs -- with the staged
backward and the two-bucket all-reduce; their averaged gradient and the parameters after the AdamW step must equal those of a single process trained on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(case="g2_32_m2", batch=2):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import types
    from recipe import CASES, LOSS_CFG, make_inputs
    from veloxseg_amd import functional as VF
    from veloxseg_amd.engine import TrainEngine
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils.loss import Loss
    cfg_d, _ = CASES[case]                            # all dropout p = 0 -> deterministic
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg_d["in_ch"]))
    x, lab = make_inputs(cfg_d, batch)
    torch.manual_seed(5)
    model = VeloxSeg(**cfg_d).cuda()
    return cfg_d, crit, x, lab, model, TrainEngine


def _worker(rank, world, port, out_dir, use_graph, backend="gloo", level_buckets=True, break_rank1=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":                 # production: one rank per GPU, RCCL over xGMI
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:                                 # the test box has one MI355X: both ranks share it over gloo
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_d, crit, x, lab, model, TrainEngine = _setup()
    # (a bucket floor of 16 KB so that this small model gets one bucket per encoder level, as the 128^3 models do at the default 1 MB)
    eng = TrainEngine(model, crit, (1, 2, 32, 32, 32), use_graph=use_graph, overlap=True, bucket_min_bytes=1 << 14, level_buckets=level_buckets)
    assert eng.world == 2 and eng.overlap
    if break_rank1 and rank == 1:         # this rank's capture "fails": it falls back to eager launches and must still issue the collectives of the taped rank
        def _boom():
            raise RuntimeError("forced capture failure (test)")
        eng._capture = _boom
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss = eng.step(x[rank:rank + 1].cuda(), lab[rank:rank + 1].cuda())
    torch.cuda.synchronize()
    if break_rank1:
        assert eng.use_graph == (rank == 0)
        use_graph = eng.use_graph
    # the bucketed all-reduces of one step tile the flat gradient buffer exactly once (tail first): eager = multi-grad hooks inside backward(),
    # tape = markers recorded by the encoder-backward tape (engine._mark) that the communication stream waits for
    assert eng.use_graph == use_graph
    cover = sorted(eng._reduced)
    assert cover[0][0] == 0 and cover[-1][1] == eng.flat.numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), eng._reduced
    assert [r[1] for r in eng._reduced] == sorted((r[1] for r in eng._reduced), reverse=True), ("buckets must be reduced tail first", eng._reduced)
    if break_rank1:
        assert len(eng._reduced) == 2, eng._reduced
    elif level_buckets or not use_graph:
        assert len(eng._reduced) >= 3, ("expected the decoder bucket and at least two encoder buckets", eng._reduced, eng.flat.plan(1 << 14))
    else:              # the taped default: the decoder bucket during the encoder backward, the encoder's gradients in one bucket after it
        assert len(eng._reduced) == 2, eng._reduced
    if rank == 0:      # flat.grad now holds the SUM over ranks (AdamW applies the 1/world scale)
        torch.save({"grad": (eng.flat.grad / world).cpu(), "param": eng.flat.param.cpu(), "loss": float(loss), "plan": eng.flat.plan()}, os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X (RCCL); the one-GPU box runs the gloo variant below")
def test_two_rank_engine_over_rccl_equals_single_process(tmp_path):
    """the same check on the production backend: torch.distributed "nccl" = RCCL, one rank per GPU"""
    _check_two_ranks(tmp_path, False, "nccl")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("use_graph,level_buckets,break_rank1", [(False, True, False), (True, True, False), (True, False, False), (True, False, True)],
                         ids=["eager", "tape_level_buckets", "tape_default", "tape_default_rank1_falls_back_to_eager"])
def test_two_rank_engine_equals_single_process(tmp_path, use_graph, level_buckets, break_rank1):
    _check_two_ranks(tmp_path, use_graph, "gloo", level_buckets, break_rank1)


def _check_two_ranks(tmp_path, use_graph, backend, level_buckets=True, break_rank1=False):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), use_graph, backend, level_buckets, break_rank1), nprocs=world, join=True)
    dp = torch.load(os.path.join(str(tmp_path), "dp.pt"))
    cfg_d, crit, x, lab, model, TrainEngine = _setup()
    eng = TrainEngine(model, crit, (2, 2, 32, 32, 32), use_graph=False, overlap=True)    # world 1: overlap off automatically
    eng.step(x.cuda(), lab.cuda())
    torch.cuda.synchronize()
    g_ref = eng.flat.grad.cpu()
    rel = float((dp["grad"] - g_ref).norm() / g_ref.norm())
    assert rel < 1e-5, f"averaged 2-rank gradient differs from the global-batch gradient: rel {rel:.3e}"
    # parameters after one AdamW step: identical wherever the gradient is not round-off noise (|g| > 1e-6: Adam's first step is sign-like)
    mask = g_ref.abs() > 1e-6
    err = float((dp["param"] - eng.flat.param.cpu())[mask].abs().max())
    assert err < 1e-6, f"parameters after one data-parallel AdamW step differ by {err:.3e}"


@pytest.mark.timeout(900)
def test_bench_two_ranks_end_to_end_over_gloo():
    """`python bench.py --gpus 2 --backend gloo`: the driver's command line for N > 1 (self-launched ranks, rendezvous on 127.0.0.1, barrier + max-over-ranks timing,
    rank 0 prints ONE parsable JSON line) end to end -- both ranks share the test box's one MI355X over gloo; production is --backend nccl (RCCL), one rank per GPU."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--workload", "autopet96", "--batch", "1",
                        "--no-cpu-baseline", "--no-eager-baseline", "--no-kernel-pass", "--dispersion-steps", "0"], capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, ("exactly one JSON line from rank 0", r.stdout[-1000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["global_batch"] == 2 and d["config"]["parallelism"] == "dp2"
    # the self-diagnosing communication block: every field the first RCCL run will be read by
    c = d["comm"]
    assert c["rccl_ranks"] == 2 and c["placement"] == "lane" and c["step_ms_no_comm"] > 0 and isinstance(c["exposed_ms"], float)
    assert sum(b["bytes"] for b in c["buckets"]) == c["payload_bytes_per_step"] and len(c["buckets"]) == 2, c["buckets"]
    assert all(b["count"] == c["steps_sampled"] and b["ms_mean"] > 0 and b["stream"] for b in c["buckets"])
    assert c["buckets"][0]["stream"].startswith("tape lane"), c["buckets"]         # the decoder bucket rides the dec_wg lane (engine.comm_placement = "lane")


def _worker_big(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_d, crit, x, lab, model, TrainEngine = _setup("g6_128_brats", 4 * world)
    eng = TrainEngine(model, crit, (4, sum(cfg_d["in_ch"]), *cfg_d["input_size"]), lr=0.0, weight_decay=0.0, use_graph=True, overlap=True)
    eng.comm_profile = []
    eng.step(x[4 * rank:4 * rank + 4].cuda(), lab[4 * rank:4 * rank + 4].cuda())
    torch.cuda.synchronize()
    assert eng.use_graph and eng.graphs is not None
    eng.step()                                   # a replayed step (lr = 0: same parameters)
    torch.cuda.synchronize()
    rep = eng.comm_report()
    assert len(rep) == 2 and sum(b["bytes"] for b in rep) == eng.flat.numel * 4, rep
    if rank == 0:
        torch.save({"grad": (eng.flat.grad / world).cpu(), "loss": float(eng.loss)}, os.path.join(out_dir, "dp_big.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_two_rank_taped_engine_at_the_per_gpu_shape_of_baseline_config_3(tmp_path):
    """BASELINE configs[3]'s per-GPU shape -- brats128, B = 4 per rank -- on two ranks (gloo: both share the box's one MI355X), taped default (decoder bucket beside the
    encoder backward, encoder bucket after it): the averaged gradient equals the gradient of ONE process on the global batch of 8."""
    world = 2
    mp.spawn(_worker_big, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    dp = torch.load(os.path.join(str(tmp_path), "dp_big.pt"))
    cfg_d, crit, x, lab, model, TrainEngine = _setup("g6_128_brats", 4 * world)
    eng = TrainEngine(model, crit, (8, sum(cfg_d["in_ch"]), *cfg_d["input_size"]), lr=0.0, weight_decay=0.0, use_graph=False, overlap=False)
    eng.step(x.cuda(), lab.cuda())
    torch.cuda.synchronize()
    g_ref = eng.flat.grad.cpu()
    rel = float((dp["grad"] - g_ref).norm() / g_ref.norm())
    assert rel < 2e-5, f"averaged 2-rank gradient differs from the global-batch gradient: rel {rel:.3e}"


def _world1_nccl_worker(out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res = {}
    for mode in ("plain", "nccl_eager", "nccl_tape", "nccl_tape_pipe"):
        cfg_d, crit, x, lab, model, TrainEngine = _setup(batch=2)
        eng = TrainEngine(model, crit, (2, 2, 32, 32, 32), use_graph=mode != "nccl_eager", overlap=True, bucket_min_bytes=1 << 14,
                          force_comm=mode != "plain", pipeline_tail=mode == "nccl_tape_pipe")
        assert eng.dp == (mode != "plain")
        losses = []
        for it in range(3):
            losses.append(float(eng.step(x.cuda(), lab.cuda())))
            eng.flush()
            torch.cuda.synchronize()
            if it == 0:
                g0 = eng.flat.grad.detach().cpu().clone()           # (the gradient of the first step: same parameters in every mode)
        if mode != "plain":
            cover = sorted(eng._reduced)
            assert cover[0][0] == 0 and cover[-1][1] == eng.flat.numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), eng._reduced
        res[mode] = (losses, g0, bool(eng.use_graph))
        del eng, model
    dist.destroy_process_group()
    torch.save(res, os.path.join(out_dir, "w1.pt"))


@pytest.mark.timeout(900)
def test_world_size_one_rccl_group_runs_the_real_collective_path(tmp_path):
    """torch.distributed "nccl" (= RCCL) with ONE rank on the one MI355X of the test box.  TrainEngine(force_comm=True) then issues its bucketed
    all-reduces through ProcessGroupNCCL's own stream / events -- eager hooks, the taped step (decoder bucket on the dec_wg lane, encoder bucket after the tape) and
    the pipelined tail -- and the first step's loss and gradient must equal those of the same engine without a process group (a one-rank sum changes nothing), three steps' losses must track."""
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_world1_nccl_worker, args=(str(tmp_path),))
    p.start()
    p.join(800)
    assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    res = torch.load(os.path.join(str(tmp_path), "w1.pt"))
    ref_losses, ref_g, _ = res["plain"]
    for mode in ("nccl_eager", "nccl_tape", "nccl_tape_pipe"):
        losses, g0, graph = res[mode]
        assert graph == (mode != "nccl_eager"), mode
        # first step: same parameters, so loss and gradient agree to summation-order noise; later steps go through Adam's normalised update, which turns the noise of
        # near-zero gradient elements into +-lr differences of single parameters: the losses still track
        assert abs(losses[0] - ref_losses[0]) <= 1e-5 * abs(ref_losses[0]), (mode, losses, ref_losses)
        assert float((g0 - ref_g).abs().max()) <= 2e-5 * float(ref_g.abs().max()), (mode, float((g0 - ref_g).abs().max()), float(ref_g.abs().max()))
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= 2e-3 * abs(b), (mode, losses, ref_losses)
