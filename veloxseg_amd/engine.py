"""Training-step engine for MI355X: flat parameter / gradient buffers, hipGraph-captured forward+loss+backward,
data-parallel gradient all-reduce over RCCL (torch.distributed backend "nccl"), fused AdamW on the flat buffer.

Reference step loop being reproduced: utils/train_brats2021.py:225-241 (zero_grad -> model -> Loss -> backward ->
optimizer.step) with AdamW(lr 2.5e-4, wd 0.01) from config/train_config_bs4.json:66-72.  The reference is single
process / single device; data parallelism by 3-D patch is new here: every op is per-sample, so the
average of per-rank gradients equals the gradient of the global batch exactly when per-rank batches are equal.

Buckets: parameters are laid out [encoder | decoders] in ONE flat fp32 buffer; gradients likewise.  The decoder
gradients are complete first (backward runs decoders -> encoder), so with `overlap=True` the backward is split at
the encoder outputs: the decoder bucket's all-reduce runs on a side stream while the encoder backward executes.
"""
from __future__ import annotations

import contextlib
import os
import warnings
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _hip as H
from . import functional as VF

TAPE_WGRAD_SIDE = os.environ.get("VELOXSEG_TAPE_WGRAD_SIDE", "0") != "0"
# the flat-gradient fill at the tail of the segmentation decoder's forward lane (default) or at the head of the encoder forward (0; A/B)
ZERO_GRAD_IN_DEC_FWD = os.environ.get("VELOXSEG_ZERO_GRAD_DEC", "1") != "0"
# the weight images of the JLC blocks built ahead, on a side lane at the head of the encoder forward (0: in front of every block, A/B)
WIMG_PREFETCH = os.environ.get("VELOXSEG_WIMG_PREFETCH", "1") != "0"
# ... and those of the patch-expand heads: OFF.  Same box, with / without: autopet128 +0.1 ... 0.3 %, autopet96 +0.4 %, hecktor +0.6 %, but brats128 fp32 -4 % (755 -> 723 patches/s):
# its two decoder lanes then reach their 100 us patch-expand kernels at the same moment and the decoder fan takes 576 instead of 458 us although its critical path is 15 us
# shorter (tools/tape_critical_path.py brats128) -- the two preparation launches had been staggering them.
EXPAND_PREFETCH = os.environ.get("VELOXSEG_EXPAND_PREFETCH", "0") == "1"
# TapedPredictor: the weight images of the JLC blocks built once and kept across forwards (0: rebuilt inside every replay, A/B)
PREDICTOR_KEEP_IMAGES = os.environ.get("VELOXSEG_PREDICTOR_KEEP_IMAGES", "1") != "0"
TAPE_PGO = os.environ.get("VELOXSEG_TAPE_PGO", "0") == "1"                      # profile-guided lane layout of the encoder tapes (csrc/tape.hip vx_tape_build_pgo)
TAPE_PGO_STAGES = tuple(k for k in os.environ.get("VELOXSEG_TAPE_PGO_STAGES", "enc_bwd,enc_fwd").split(",") if k)
TAPE_WGRAD_DEFER = os.environ.get("VELOXSEG_TAPE_WGRAD_DEFER", "1") != "0"      # taped encoder backward: weight gradients at the end of their own stream
WG_EARLY = os.environ.get("VELOXSEG_WG_EARLY", "1") != "0"      # taped steps: dec_wg[k] queued behind dec_bwd[k] instead of behind the join of the decoder-backward fan
WGRAD_STREAM = os.environ.get("VELOXSEG_WGRAD_STREAM", "0") != "0"      # experiment (off): weight-gradient kernels deferred to a side stream -- measured 13.5 vs 12.0 ms/step, they steal CUs from the critical path


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def param_bucket(name: str) -> int:
    """Bucket of a parameter = the order in which backward completes its gradient, LAST first: 0..3 = encoder levels 1..4 (PatchEmbed with level 1;
    `encoder_attn.layers.k` incl. its PatchMerging, `encoder_conv.down{k+1} / layer{k+1}`, `attn2conv_{k+1}` with level k+1), 4 = the decoders.
    Backward runs decoders -> level 4 -> ... -> level 1, so the flat buffer [L1 | L2 | L3 | L4 | decoders] is reduced from its tail to its head."""
    import re as _re
    if not name.startswith("encoder."):
        return 4
    for pat, off in ((r"encoder\.encoder_attn\.layers\.(\d+)\.", 0), (r"encoder\.encoder_conv\.(?:down|layer)(\d+)", -1), (r"encoder\.attn2conv_(\d+)", -1)):
        m = _re.match(pat, name)
        if m:
            return min(max(int(m.group(1)) + off, 0), 3)
    return 0


class FlatParams:
    """Re-homes every parameter (and its .grad) of `model` as a view into one flat fp32 buffer, grouped by bucket (param_bucket)."""

    def __init__(self, model: torch.nn.Module, bucket_of=param_bucket):
        params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0][1].device
        order = sorted(range(len(params)), key=lambda i: (bucket_of(params[i][0]), i))      # stable: registration order inside a bucket
        self.names: List[str] = []
        self.slices = {}
        self.bounds = {}                     # bucket -> (lo, hi) element range of the flat buffers
        off = 0
        for i in order:
            n, p = params[i]
            b = bucket_of(n)
            if b not in self.bounds:
                self.bounds[b] = [off, off]
            self.slices[n] = (off, p.numel())
            self.names.append(n)
            off = _align(off + p.numel())
            self.bounds[b][1] = off
        self.bounds = {b: tuple(v) for b, v in self.bounds.items()}
        self.split = self.bounds[4][0] if 4 in self.bounds else off          # start of the decoder bucket
        self.numel = off
        self.param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for i in order:
                n, p = params[i]
                o, k = self.slices[n]
                self.param[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.param[o:o + k].view(p.shape)
                p.grad = self.grad[o:o + k].view(p.shape)
        self.params = [params[i][1] for i in order]

    def plan(self, min_bytes: int = 1 << 20):
        """All-reduce plan for the overlapped backward: [(trigger, lo, hi)], tail first.  trigger 4 = "decoders done", 3 / 2 / 1 = "encoder level
        4 / 3 / 2 done" (multi-grad hooks), 0 = after backward.  Buckets smaller than min_bytes are merged into the NEXT (later-finishing) one, since
        at this payload a collective is latency-bound."""
        out, lo_pending = [], None
        for b in (4, 3, 2, 1, 0):
            if b not in self.bounds:
                continue
            lo, hi = self.bounds[b]
            if lo_pending is not None:
                hi = lo_pending[1]
            if b != 0 and b != 4 and (hi - lo) * 4 < min_bytes:
                # too small: reduce together with the next bucket down.  (Never the decoder bucket: the taped step always reduces [split, n) on its
                # own between the decoder and the encoder backward, and a rank that fell back to eager launches must issue the same collectives.)
                lo_pending = (lo, hi)
                continue
            out.append((b, lo, hi))
            lo_pending = None
        if lo_pending is not None:
            out.append((0, lo_pending[0], lo_pending[1]))
        return out

    def taped_schedule(self, min_bytes: int = 1 << 20, markers=()):
        """The all-reduce slices of ONE taped data-parallel step, in issue order: [(trigger, lo, hi)].  First the decoder bucket [split, numel) (trigger 4: reduced
        beside the encoder backward), then -- only for the levels whose marker the encoder-backward tape carries (`markers`, opt-in per-level buckets) -- one slice
        per plan entry, tail first, and last whatever is left of the encoder [0, done) (trigger 0).  The slices tile [0, numel) exactly once for every `markers`
        subset; TrainEngine._replay issues exactly this list, and a rank that fell back to eager launches must issue the same one (tests/test_dp_gloo_cpu.py)."""
        out = [(4, self.split, self.numel)]
        done = self.split
        for trig, lo, hi in self.plan(min_bytes):            # tail first: (decoders), level 4, level 3, ...
            if trig == 4:
                continue
            hi = min(hi, done)            # a small decoder bucket is merged into the next plan entry: its tail [split, numel) is already reduced
            if hi <= lo:
                continue
            if trig in markers:
                out.append((trig, lo, hi))
                done = lo
        if done > 0:
            out.append((0, 0, done))
        return out

    def zero_grad(self):
        self.grad.zero_()

    def reattach(self):
        """make sure .grad still aliases the flat buffer (e.g. after zero_grad(set_to_none=True))"""
        for n, p in zip(self.names, self.params):
            o, k = self.slices[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + k].view(p.shape)


class LaunchTape:
    """A captured stage (torch.cuda.CUDAGraph kept as a raw hipGraph_t, never instantiated) read back once and replayed as plain launches on
    several HIP streams by csrc/tape.hip.  Holds the graph: its nodes own the kernel arguments and its pool owns the memory."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", max_lanes: int = 6):
        import ctypes
        self.graph = graph
        h = ctypes.c_void_p()
        H.call("vx_tape_build", graph.raw_cuda_graph(), int(max_lanes), ctypes.addressof(h))
        self.handle = h.value
        info = (ctypes.c_int * 4)()
        base = ctypes.addressof(info)
        H.call("vx_tape_info", self.handle, base, base + 4, base + 8, base + 12)
        self.n_nodes, self.n_kernels, self.n_lanes, self.n_events = (int(v) for v in info)

    def replay(self):
        H.call("vx_tape_replay", self.handle, H.stream_ptr())

    def set_lane_rotation(self, k: int):
        """lane l of this tape on pool stream (l + k) % 4: a tape replayed BESIDE another one (csrc/tape.hip vx_tape_set_lane_rotation)"""
        H.call("vx_tape_set_lane_rotation", self.handle, int(k) % 4)

    def relayout(self, max_lanes: int):
        """profile-guided layout (csrc/tape.hip vx_tape_build_pgo): every node is timed alone (its buffers must hold the values of a real replay), then
        the graph is laid out again by list scheduling with those durations -- longest remaining path first, each node on the lane where it can start
        earliest -- instead of greedily in capture order"""
        import ctypes
        n = self.n_nodes
        if n < 4 or self.n_lanes < 2:
            return
        us = (ctypes.c_float * n)()
        H.call("vx_tape_profile", self.handle, H.stream_ptr(), 2, ctypes.addressof(us))
        h = ctypes.c_void_p()
        H.call("vx_tape_build_pgo", self.graph.raw_cuda_graph(), int(max_lanes), ctypes.addressof(us), n, ctypes.addressof(h))
        old, self.handle = self.handle, h.value
        H.call("vx_tape_free", old)
        info = (ctypes.c_int * 4)()
        base = ctypes.addressof(info)
        H.call("vx_tape_info", self.handle, base, base + 4, base + 8, base + 12)
        self.n_nodes, self.n_kernels, self.n_lanes, self.n_events = (int(v) for v in info)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                H.call("vx_tape_free", h)
            except Exception:
                pass


class _EmptyTape:
    """a captured stage without a single launch: nothing to build, nothing to replay"""
    n_nodes = n_kernels = n_lanes = n_events = 0
    handle = None

    def replay(self):
        pass


class TapedPredictor:
    """An eval-mode forward of a fixed input shape captured once and replayed as a launch tape: the `predictor` of
    utils.inference_runtime.sliding_window_inference (the reference runs `model(window_batch)` through autograd-free eager launches,
    utils/inference_brats.py:41-53,209-216; here the ~250 launches of a window batch cost the host ~0.6 ms instead of ~2 ms and the modality /
    conv-chain branches run on their own hardware queues).  Inputs of another shape (the ragged last batch) take the eager forward.
    The first call of a shape captures and checks the replay against the eager forward; a mismatch disables the tape for that shape."""

    def __init__(self, model: torch.nn.Module, max_lanes: int = 6, check: bool = True):
        self.model, self.max_lanes, self.check = model, int(max_lanes), bool(check)
        self._tapes = {}

    def _first(self, out):
        return out[0] if isinstance(out, (list, tuple)) else out

    def _capture(self, x):
        # outside inference mode (no_grad instead): torch.cuda.graph registers the default generator's state tensors at capture_begin, and tensors
        # born in inference mode there make every later capture of the process fail ("inplace update to inference tensor")
        with torch.inference_mode(False), torch.no_grad():
            return self._capture_body(x)

    def _keep_weight_images(self, x):
        """(round 6) between the forwards of a predictor the weights do not change: the weight images of the JLC blocks the eval forward runs (encoder conv branch, segmentation
        decoder) are built ONCE, before the capture, and kept (functional.jlc_prefetch(keep=True)) -- the tape then holds none of their preparation launches (14 of ~250).
        They are valid while the signature below holds (storage, version counters, the weights epoch a TrainEngine bumps); _vxops drops an entry that is not."""
        if not PREDICTOR_KEEP_IMAGES:
            return
        from .model.components.conv_blocks import JLC
        mdl = getattr(self.model, "_model", self.model)        # (utils.inference_runtime.Net wraps the network)
        enc = getattr(mdl, "encoder", None)
        ce = getattr(enc, "encoder_conv", None)
        dec = getattr(mdl, "decoder", None)
        if ce is None or dec is None or x.dim() != 5 or not hasattr(mdl, "patch_size"):
            return
        ps = mdl.patch_size
        ps = [int(ps)] * 3 if not isinstance(ps, (tuple, list)) else [int(v) for v in ps]
        base = [int(s_) // p_ for s_, p_ in zip(x.shape[2:], ps)]
        grid = lambda lvl: [max(g // (2 ** (lvl - 1)), 1) for g in base]
        cur = torch.cuda.current_stream(x.device)
        for i in (1, 2, 3, 4):
            for blk in getattr(ce, f"layer{i}"):
                if isinstance(blk, JLC):
                    VF.jlc_prefetch(blk, grid(i), cur, keep=True)
        for lvl in (3, 2, 1):
            for blk in getattr(dec, f"layer{lvl}", []):
                if isinstance(blk, JLC):
                    VF.jlc_prefetch(blk, grid(lvl), cur, keep=True)
        head = getattr(dec, "out_conv1", None)
        if head is not None:                             # the patch-expand head's forward image too (one decoder: no second lane to collide with, cf. EXPAND_PREFETCH)
            VF.expand_prefetch(head[0], cur, keep=True)

    def _capture_body(self, x):
        dev = x.device
        xs = torch.empty(x.shape, dtype=x.dtype, device=dev)
        xs.copy_(x)
        self._keep_weight_images(xs)
        prev_ms = VF.MODALITY_STREAMS
        VF.MODALITY_STREAMS = max(VF.MODALITY_STREAMS, 2)      # (for the capture only: the setting is baked into the tape, the process-wide value is restored)
        try:
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                ref = None
                for _ in range(2):                      # lazy initialisation and allocator warm-up outside the capture
                    ref = self._first(self.model(xs)).float().clone()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._first(self.model(xs))
        finally:
            VF.MODALITY_STREAMS = prev_ms
        tape = LaunchTape(g, self.max_lanes)
        if self.check:
            tape.replay()
            torch.cuda.synchronize(dev)
            if not torch.allclose(out.float(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max()))):
                warnings.warn("TapedPredictor: the replay does not reproduce the eager forward; this shape keeps eager launches")
                return None
        return (xs, out, tape)

    def _signature(self):
        """where the parameters and buffers live: a tape holds raw device pointers, so it is only valid while nothing was re-homed (TrainEngine's
        flat buffer sets p.data, model.to() / .float(), a swapped parameter).  The module tree is walked ONCE into (owner dict, name) slots; each call
        then costs two dict look-ups per tensor and still sees a swapped Parameter object."""
        slots = getattr(self, "_sig_slots", None)
        if slots is None:
            slots = self._sig_slots = ([(m._parameters, n) for m in self.model.modules() for n, p in m._parameters.items() if p is not None]
                                       + [(m._buffers, n) for m in self.model.modules() for n, b in m._buffers.items() if b is not None])
        out = [VF.weights_epoch() if PREDICTOR_KEEP_IMAGES else 0]
        for d, n in slots:
            t = d.get(n)
            # (_version: an in-place update of a weight -- optimizer.step(), load_state_dict -- invalidates the weight images the tapes were captured without)
            out.append(None if t is None else (t.data_ptr(), t.dtype, t.device.index, t._version if PREDICTOR_KEEP_IMAGES else 0))
        return tuple(out)

    # a tape holds a CUDAGraph and raw handles: copies / pickles of the model carry a FRESH, empty predictor (copy.deepcopy(model) for an EMA copy,
    # torch.save(model)); a copied handle would also be freed twice
    def __deepcopy__(self, memo):
        import copy
        return TapedPredictor(copy.deepcopy(self.model, memo), self.max_lanes, self.check)

    def __reduce__(self):
        return (TapedPredictor, (self.model, self.max_lanes, self.check))

    @torch.inference_mode()
    def __call__(self, x: torch.Tensor):
        if self.model.training or not x.is_cuda:
            return self.model(x)
        sig = self._signature()
        if sig != getattr(self, "_sig", None):
            self._tapes, self._sig = {}, sig          # the storage moved (or the weights changed) since the capture: drop every tape of the old addresses / old weight images
        key = (tuple(x.shape), x.dtype, str(x.device))
        if key not in self._tapes:
            try:
                self._tapes[key] = self._capture(x.contiguous())
            except Exception as e:                 # a capture that cannot be taken must not take the inference run down
                warnings.warn(f"TapedPredictor: capture failed ({type(e).__name__}: {str(e)[:200]}); eager launches for shape {tuple(x.shape)}")
                self._tapes[key] = None
                torch.cuda.synchronize(x.device)
        ent = self._tapes[key]
        if ent is None:
            return self.model(x)
        xs, out, tape = ent
        xs.copy_(x, non_blocking=True)
        tape.replay()
        return out

    @torch.inference_mode()
    def forward_slot(self, x: torch.Tensor, slot: int):
        """(round 6) the same forward from one of several REPLICAS of the tape (own static input / output, lanes rotated by 2 per slot): two window batches of a sliding-window
        inference in flight at once, each replayed from its own caller stream (utils.inference_runtime.sliding_window_inference).  Returns None where this shape has no tape
        (the caller then calls the predictor the plain way)."""
        if slot <= 0:
            return self(x)
        if self.model.training or not x.is_cuda:
            return None
        sig = self._signature()
        if sig != getattr(self, "_sig", None):
            return None                                     # (slot 0 notices first and drops the tapes; until then the plain call)
        key = (tuple(x.shape), x.dtype, str(x.device))
        if self._tapes.get(key) is None:
            return None
        rk = (key, int(slot))
        reps = self.__dict__.setdefault("_replicas", {})
        if reps.get("sig") != sig:
            reps.clear()
            reps["sig"] = sig
        if rk not in reps:
            try:
                ent = self._capture(x.contiguous())
                if ent is not None:
                    ent[2].set_lane_rotation({1: 2, 2: 0, 3: 1, 4: 3}.get(int(slot), int(slot)))      # (replica 1 beside the plain tape: its main chains on the other two queues)
                reps[rk] = ent
            except Exception as e:
                warnings.warn(f"TapedPredictor: capture of replica {slot} failed ({type(e).__name__}: {str(e)[:200]})")
                reps[rk] = None
                torch.cuda.synchronize(x.device)
        ent = reps[rk]
        if ent is None:
            return None
        xs, out, tape = ent
        xs.copy_(x, non_blocking=True)
        tape.replay()
        return out


class TrainEngine:
    """step(x, labels) = zero_grad -> forward -> loss -> backward -> [all-reduce] -> AdamW, on static buffers."""

    @property
    def last_outputs(self):
        """the training outputs of the last step in the reference's list layout (what per-step metrics read); with the staged loss the
        reconstruction channels are concatenated only when somebody asks"""
        if self._last_outputs is None and getattr(self, "_last_parts", None) is not None:
            self._last_outputs = self.model.assemble_train(self._last_parts)
        return self._last_outputs

    @last_outputs.setter
    def last_outputs(self, v):
        self._last_outputs = v
        self._last_parts = None

    def __init__(self, model, criterion, batch_shape, label_dtype=None, lr=2.5e-4, weight_decay=0.01, betas=(0.9, 0.999),
                 eps=1e-8, use_graph=False, overlap=True, process_group=None, warmup_steps=2, verify_replays=3, optimizer=None, fuse_ds=True,
                 replay="tape", tape_lanes=6, precision="fp32", bucket_min_bytes=1 << 20, level_buckets=False, pipeline_tail=False, force_comm=False):
        self.model, self.criterion = model, criterion
        # taped data-parallel steps: True = one all-reduce bucket per encoder level, released by markers inside the encoder-backward tape; False
        # (default) = the decoder bucket during the encoder backward, the encoder's gradients in one bucket after it.  A marker joins every forked
        # stream of the stage: measured on one GPU the markers cost 0.85 ms per step (7.50 vs 6.65 ms), more than the ~0.2 ms of all-reduce
        # (5.5 MB over xGMI) they can hide
        self.level_buckets = bool(level_buckets) or os.environ.get("VELOXSEG_LEVEL_BUCKETS") == "1"
        # Taped steps only: the decoders' weight gradients (a ~1 ms sink on the fourth lane), their all-reduce and the decoder half of AdamW are NOT joined at the end
        # of the step: the next step's encoder forward -- which reads encoder parameters only, and leaves a lane idle -- starts as soon as the encoder half of AdamW is
        # done, and the join moves to just before the next decoder forward.  Same arithmetic, same collectives in the same order; what changes for the caller:
        # after step() returns, work on the CURRENT stream is ordered behind the encoder update only -- read decoder parameters / gradients after flush() (or a
        # device synchronise).  Opt-in (bench.py and the step loop of utils/train_loop.py turn it on).
        self.pipeline_tail = bool(pipeline_tail) or os.environ.get("VELOXSEG_PIPELINE_TAIL") == "1"
        self._tail_pending = False
        self._two_buckets = bool(use_graph) and replay == "tape" and not self.level_buckets
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        # "bf16": bf16 MFMA operands in the patch-expand layers (functional.set_precision).  A property of THIS engine: the process-wide switches it needs
        # (precision, per-modality forks, in-place RNG step) are set for the duration of its own eager passes / captures only (`_settings`) -- a tape has them
        # baked in -- so two engines with different settings in one process do not interfere
        self.precision = precision
        self.bucket_min_bytes = int(bucket_min_bytes)      # all-reduce buckets below this size are merged into the next one (latency-bound collectives)
        if hasattr(model, "ds_fused"):
            # deep-supervision heads stay on their grids; the loss kernels interpolate (csrc/loss_ds.hip).  Row widths the kernels do not tile
            # (W/4 must divide 64: 96^3 patches) keep the reference's up-sampled output list
            # Only with the library's own Loss: any other criterion gets the reference's output list (every head at the input size).
            from .utils.loss import Loss as _Loss
            ok = (bool(fuse_ds) and isinstance(criterion, _Loss)
                  and bool(H.query("vx_seg_loss_ds_ok", int(model.n_classes), *[int(v) for v in batch_shape[2:]])))
            self._ds_fused = ok            # applied for the duration of this engine's own passes only (`_settings`), like the other switches
        self.dev = next(model.parameters()).device
        self.flat = FlatParams(model)
        self.m = torch.zeros_like(self.flat.param)
        self.v = torch.zeros_like(self.flat.param)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0
        self.optimizer = None
        if optimizer is not None:
            self.bind_optimizer(optimizer)
        self.pg = process_group
        self.skip_comm = False            # diagnostics: issue no collective (all ranks alike) -- the step time without communication
        self.comm_profile = None          # diagnostics: a list -> every all-reduce appends its HIP events (comm_report())
        # where the decoder bucket's all-reduce is enqueued in the taped step (profiles/r03_comm_standin_probe.txt measured the three on one GPU with a stand-in
        # kernel): "lane" (default) = on the dec_wg lane's stream after the encoder-backward tape; "fresh_after" = on the engine's comm stream after the tape;
        # "fresh_before" = on the comm stream BEFORE the tape (its wait for the dec_wg lane then sits in a hardware queue the tape shares)
        self.comm_placement = os.environ.get("VELOXSEG_COMM_PLACEMENT", "lane")
        if self.comm_placement not in ("lane", "fresh_after", "fresh_before"):
            raise ValueError("VELOXSEG_COMM_PLACEMENT must be lane, fresh_after or fresh_before")
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # `dp`: the step issues its collectives.  Normally world > 1; a process group of ONE rank (VELOXSEG_FORCE_COMM=1 / force_comm) runs the same code path -- the
        # all-reduces go through ProcessGroupNCCL's own stream and events for real and change no value -- which is how the RCCL path is exercised and its cost beside
        # the four busy hardware queues measured on a one-GPU box (tests/test_dp_gpu.py, profiles/r05_comm_world1_nccl.json)
        self.dp = self.world > 1 or ((bool(force_comm) or os.environ.get("VELOXSEG_FORCE_COMM") == "1") and dist.is_available() and dist.is_initialized())
        self.overlap = overlap and self.dp
        self.use_graph = use_graph
        if replay not in ("tape", "graph"):
            raise ValueError("replay must be 'tape' (csrc/tape.hip launches) or 'graph' (hipGraphLaunch)")
        self.replay_mode, self.tape_lanes = replay, int(tape_lanes)
        self._setup_branch_loss()
        B = batch_shape[0]
        self.x = torch.zeros(batch_shape, device=self.dev, dtype=torch.float32)
        if label_dtype is None:
            # the engine's own copy of the label volume: step() converts whatever the loader hands over (int64 in the reference: utils/train_autopet.py:239) while
            # it copies.  One byte per voxel when the library's loss kernels read it and the class count allows: at 128^3 x 4 the int64 labels were HALF of the
            # bytes the loss forward reads (67 of 137 MB) and a third of its backward's (VELOXSEG_LABELS=int64 for the A/B)
            from .utils.loss import Loss as _Loss2
            small = isinstance(criterion, _Loss2) and int(getattr(model, "n_classes", 1 << 30)) <= 255 and os.environ.get("VELOXSEG_LABELS", "u8") != "int64"
            label_dtype = torch.uint8 if small else torch.int64
        self.labels = torch.zeros((B, 1, *batch_shape[2:]), device=self.dev, dtype=label_dtype)
        self.loss = torch.zeros((), device=self.dev, dtype=torch.float32)
        self.graphs = None
        self.last_outputs = None
        self.comm_stream = torch.cuda.Stream(device=self.dev) if self.dp else None
        self._warm = warmup_steps
        self.verify_replays = verify_replays
        if self.dp:
            dist.broadcast(self.flat.param, src=0, group=self.pg)      # identical replicas at start
            # Data parallel: the start gate of step N+1's tapes is set behind AdamW(N), which waits for all-reduce(N) -- i.e. for the slowest PEER (data loader,
            # rank-0 checkpoint, capture skew).  Collective latency must not count against the cross-lane poll timeout (a poll that gives up lets the kernels
            # behind it run on an unmet dependency); the process group's own timeout / RCCL watchdog bounds a peer that never arrives.
            # (not "never" -- a lost cross-lane flag must still surface through vx_tape_flag_timeouts / _check_flag_timeouts instead of spinning for ever,
            # gloo has no watchdog; the user's VELOXSEG_TAPE_FLAG_TIMEOUT_MS wins, and engines that replay no tapes leave the process-wide setting alone)
            if use_graph and replay == "tape" and not os.environ.get("VELOXSEG_TAPE_FLAG_TIMEOUT_MS"):
                pg_s = None
                try:
                    pg_s = dist.distributed_c10d._get_default_timeout(dist.get_backend(process_group)).total_seconds()
                except Exception:
                    pass
                H.call("vx_tape_set_flag_timeout_ms", int(min(max(pg_s or 1800.0, 60.0), 7200.0) * 1000))

    @contextlib.contextmanager
    def _settings(self, capture: bool = False):
        """this engine's values of the process-wide switches of veloxseg_amd.functional, for the duration of one of its eager passes or captures"""
        prev = (VF.get_precision(), VF.MODALITY_STREAMS, VF.RNG_INPLACE, getattr(self.model, "ds_fused", None))
        VF.set_precision(self.precision)
        if prev[3] is not None and hasattr(self, "_ds_fused"):
            self.model.ds_fused = self._ds_fused
        if capture:
            if self.replay_mode == "tape":
                VF.MODALITY_STREAMS = max(VF.MODALITY_STREAMS, 2)      # forks cost the tape nothing on the host: every per-modality piece gets its own branch
            VF.RNG_INPLACE = True       # replays must bump the tensor the captured kernels point at
        try:
            yield
        finally:
            VF.set_precision(prev[0])
            VF.MODALITY_STREAMS, VF.RNG_INPLACE = prev[1], prev[2]
            if prev[3] is not None:
                self.model.ds_fused = prev[3]

    # ---- pieces ---------------------------------------------------------------------------------
    def _forward_loss(self):
        outs = self.model(self.x)
        return outs, self.criterion(outs, self.labels, sr_labels=self.x)

    def _fwd_bwd_single(self):
        """plain eager step: model() runs the decoder branches on forked streams (functional.run_branches)"""
        self.flat.zero_grad()
        outs, loss = self._forward_loss()
        self._backward(loss)
        self.loss.copy_(loss.detach())
        self.last_outputs = [o.detach() for o in outs]

    def _backward(self, loss):
        """loss.backward() with the weight-gradient kernels of the convolutions on a side stream (csrc/_vxops.cpp WgradSide): only the input
        gradients stay on the critical path; the launching stream is re-joined before anything reads the parameter gradients"""
        m = VF.cpp_module() if (WGRAD_STREAM and not self.use_graph) else None
        if m is None:
            loss.backward()
            self._join_side_streams()
            return
        m.set_wgrad_stream(True)
        try:
            loss.backward()
        finally:
            m.set_wgrad_stream(False)
            m.wgrad_join(torch.cuda.current_stream(self.dev).cuda_stream, self.dev.index or 0, True)
        self._join_side_streams()

    def _join_side_streams(self, waiter=None):
        """The weight-gradient kernels write flat.grad IN PLACE on whatever stream their node ran on (decoder branches, modality branches, the
        encoder conv chain) and return None to autograd, so no AccumulateGrad node carries the dependency: make the stream that will read
        flat.grad (all-reduce, AdamW) wait for every forked stream explicitly instead of relying on the engine's end-of-backward leaf sync."""
        waiter = waiter or torch.cuda.current_stream(self.dev)
        for s_ in VF.all_side_streams(self.dev):
            waiter.wait_stream(s_)

    def _fwd_bwd_overlapped(self):
        """plain eager step + bucketed all-reduces started from INSIDE the backward pass (FlatParams.plan): the decoder bucket when the gradient of
        enc4 is complete (every decoder consumes enc4 in its first layer, so its gradient is final exactly when the three decoders have queued their
        whole backward), then one bucket per encoder level when the gradients of that level's INPUTS (the tensors level L-1 hands to level L) are
        complete; each runs on the communication stream while the earlier levels' backward executes.  Only the last, smallest bucket (levels 1-2,
        0.65 MB of the 9.2 MB payload) is reduced after backward().  No staging, no detached leaves."""
        self.flat.zero_grad()
        handles = []
        plan = self.flat.plan(self.bucket_min_bytes)
        if self._two_buckets:
            # an engine that was asked for tapes reduces [decoders | encoder] (see level_buckets); a rank whose capture failed and fell back to these eager
            # launches must issue the SAME collectives as the ranks that replay tapes
            plan = [(4, self.flat.split, self.flat.numel), (0, 0, self.flat.split)] if self.flat.split > 0 else [(0, 0, self.flat.numel)]
        self._reduced = []
        by_trigger = {t: (lo, hi) for t, lo, hi in plan}
        fired = set()

        def reduce_now(trigger):
            fired.add(trigger)
            lo, hi = by_trigger[trigger]
            cur = torch.cuda.current_stream(self.dev)
            self.comm_stream.wait_stream(cur)
            m = VF.cpp_module() if WGRAD_STREAM else None
            if m is not None:
                m.wgrad_join(self.comm_stream.cuda_stream, self.dev.index or 0, False)      # weight gradients deferred to the side stream
            for s_ in VF.all_side_streams(self.dev):      # the weight-gradient kernels ran on the streams of their nodes
                self.comm_stream.wait_stream(s_)
            with torch.cuda.stream(self.comm_stream):
                self._allreduce(lo, hi)

        def on_enc(attn, encs):
            if 4 in by_trigger:
                handles.append(torch.autograd.graph.register_multi_grad_hook([encs[-1]], lambda _g: reduce_now(4), mode="all"))

        def on_level(level, tensors):               # level = 2..4 (1-based): `tensors` are that level's inputs
            trig = level - 1                         # level 4 done -> trigger 3, ...
            ts = [t for t in tensors if t.requires_grad]
            if trig in by_trigger and ts:
                handles.append(torch.autograd.graph.register_multi_grad_hook(ts, lambda _g, trig=trig: reduce_now(trig), mode="all"))

        self.model._on_encoder_outputs = on_enc
        self.model.encoder._on_level_inputs = on_level
        try:
            outs, loss = self._forward_loss()
            self._backward(loss)
        finally:
            self.model._on_encoder_outputs = None
            self.model.encoder._on_level_inputs = None
            for h in handles:
                h.remove()
        cur = torch.cuda.current_stream(self.dev)
        for t, lo, hi in plan:                       # whatever did not fire from a hook (always trigger 0), in plan order
            if t not in fired:
                self.comm_stream.wait_stream(cur)
                with torch.cuda.stream(self.comm_stream):
                    self._allreduce(lo, hi)
        cur.wait_stream(self.comm_stream)
        self._check_tiling()
        self.loss.copy_(loss.detach())
        self.last_outputs = [o.detach() for o in outs]

    # ---- stages -------------------------------------------------------------------------------
    # The step is cut at the encoder outputs and at the decoder outputs into stages that exchange DETACHED leaves:
    #   enc_fwd -> { dec_fwd[k] } -> loss (+ its backward) -> { dec_bwd[k] } -> enc_bwd          k = 0..M (Seg decoder, M RC decoders)
    # The braces run concurrently on one HIP stream per branch.  Eagerly the same functions run back to back (stage order is
    # a valid serialisation), which is the reference the captured graphs are checked against.
    def _pipe_active(self):
        return bool(self.pipeline_tail and self.use_graph and self.replay_mode == "tape" and not self.level_buckets and self._split_dec_wgrad())

    def _s_enc_fwd(self):
        if self._pipe_active():
            self.flat.grad[:self.flat.split].zero_()       # (pipelined tail: the decoder half may still be in use by the previous step's dec_wg / AdamW: zeroed in _s_loss)
        elif not ZERO_GRAD_IN_DEC_FWD:
            self.flat.zero_grad()                          # (default: at the end of the segmentation decoder's forward lane, _s_dec_fwd -- off the head of the step's chain)
        VF.advance_rng(self.dev)
        self._prefetch_weight_images()
        self._drop_level_hooks()
        if self._mark_levels():
            # data-parallel tape: a marker where the gradients of a level's inputs are complete (multi-grad hooks, as in the eager overlapped step);
            # the encoder-backward tape records an event there and the bucket of that level is all-reduced while the lower levels still run
            by_trigger = {t: (lo, hi) for t, lo, hi in self.flat.plan(self.bucket_min_bytes)}

            def on_level(level, tensors):
                trig = level - 1
                ts = [t for t in tensors if t.requires_grad]
                if trig in by_trigger and trig != 4 and ts:
                    self._level_hooks.append(torch.autograd.graph.register_multi_grad_hook(ts, lambda _g, trig=trig: self._mark(trig), mode="all"))
            self.model.encoder._on_level_inputs = on_level
        try:
            attn, encs = self.model.encoder(self.x)
        finally:
            self.model.encoder._on_level_inputs = None
        if getattr(self, "_pf_blocks", None):
            # the side lane that built the weight images re-joins this stage (the decoders' images are complete before the decoder fan; a captured stage may not end with
            # work that never came back to its stream)
            torch.cuda.current_stream(self.dev).wait_stream(VF.side_stream(self.dev, "weight_images"))
        self._boundary = list(encs) + [t for lvl in attn for t in lvl]
        if getattr(self, "_bl", None) is not None:
            if self._bl == "pending":
                self._bl = self.criterion.staged(self._n_ds_heads())
            self._bl.begin(self.dev, self.x.shape[0], self._n_classes())
            self._rc_c = [None] * self.model.num_branches

    def _n_ds_heads(self):
        return 4           # VeloxSeg.decode_branch(0) returns pred_0 .. pred_3 + the Gram matrix (VeloxSeg.py:199-221); checked in _s_dec_fwd

    def _n_classes(self):
        return int(getattr(self.model, "n_classes", 0) or self.model.decoder.n_classes)

    def _mark_levels(self):
        return self.use_graph and self.replay_mode == "tape" and ((self.dp and self.overlap and self.level_buckets) or os.environ.get("VELOXSEG_FORCE_MARKERS") == "1")

    def _drop_level_hooks(self):
        for h in getattr(self, "_level_hooks", []):
            h.remove()
        self._level_hooks = []

    def _mark(self, trig):
        """(inside the encoder backward, on autograd's device thread) everything queued so far on the forked streams joins the STAGE's stream, then
        the marker goes there.  Not the thread's current stream: in the hook that is the NULL stream, and touching it during a capture kills the capture."""
        st = self._stage_stream
        m = VF.cpp_module() if getattr(self, "_enc_bwd_defer", False) else None
        if m is not None:
            # the weight-gradient launches of this level are deferred closures (set_wgrad_defer in _s_enc_bwd): launch them NOW, each on the stream it
            # was queued from, so that the marker -- and with it the all-reduce of this level's bucket -- is ordered after them.  Without this the
            # bucket was reduced before its deferred weight gradients ran and they landed on top of the reduced values (ranks diverge).
            m.wgrad_join(st.cuda_stream, self.dev.index or 0, True)
        with torch.cuda.stream(st):
            capturing = torch.cuda.is_current_stream_capturing()
        for s_ in VF.all_side_streams(self.dev):
            if s_.cuda_stream == st.cuda_stream:
                continue
            if capturing:
                with torch.cuda.stream(s_):
                    if not torch.cuda.is_current_stream_capturing():
                        continue                    # not part of this capture: nothing of this stage runs there
            st.wait_stream(s_)
        H.call("vx_tape_mark", int(trig), st.cuda_stream)

    def _s_dec_fwd(self, k):
        # every branch gets its own leaves (same storage, separate .grad), so concurrent branches never accumulate into one tensor
        self._prefetch_settle()
        M = self.model.num_modalities
        if self._pipe_active():
            # pipelined tail: this decoder's weight-gradient tape (dec_wg[k]) still runs while the NEXT step's encoder forward rewrites the boundary tensors, and some of
            # its kernels read them (enc2rc / up-conv / head weight gradients): the branch works on private copies of the boundary tensors it reads (~16-32 MB, copied
            # on this branch's lane at the start of its forward)
            used = set(range(4)) if k == 0 else set(range(4)) | {4 + L * M + (k - 1) for L in range(4)}
            leaves = [(t.detach().clone() if j in used else t.detach()).requires_grad_(True) for j, t in enumerate(self._boundary)]
        else:
            leaves = [t.detach().requires_grad_(True) for t in self._boundary]
        encs, flat_attn = leaves[:4], leaves[4:]
        attn = [flat_attn[L * M:(L + 1) * M] for L in range(4)]
        self._leaves[k] = leaves
        dec = self.model.decoder if k == 0 else self.model.rc_decoders[k - 1]
        dec.head_bf16 = self._head_bf16(k)          # bf16 storage mode: this branch's full-resolution output (and, from the staged loss, its gradient) as bfloat16
        try:
            self._outs[k] = list(self.model.decode_branch(k, attn, encs))
        finally:
            dec.head_bf16 = False
        bl = self._branch_loss()
        if bl is not None:        # this branch's share of the loss forward, on this branch's stream (functional.StagedLoss)
            if k == 0:
                if bl.nh != len(self._outs[0]) - 1:
                    raise RuntimeError("staged loss: unexpected number of deep-supervision heads")
                bl.seg_forward([t.detach() for t in self._outs[0][:-1]], self.labels)
            else:
                self._rc_c[k] = bl.rc_forward(self._outs[k][0].detach(), self.x, self._ch_off[k - 1])
        if k == 0 and ZERO_GRAD_IN_DEC_FWD and not self._pipe_active():
            # (round 6) the flat gradient is zeroed HERE, at the tail of the segmentation decoder's forward lane (the shortest of the fan: 439 / 452 / 451 us), instead of at the
            # head of the encoder forward: nothing accumulates into it before the backward stages, and the 9 MB fill leaves the step's critical chain
            self.flat.zero_grad()

    def _prefetch_weight_images(self):
        """(round 6) the weight images of every JLC block of the step -- functions of the weights alone -- are built at the head of the encoder forward on a side lane, in the
        order the blocks need them (encoder levels 1..4, then every decoder's levels 3, 2, 1), instead of in front of each block's convolutions on the forward chains
        (two or three 6 us launches per block: functional.jlc_prefetch).  The decoders' images are ready long before the decoder fan starts (the stage joins its side streams)."""
        if not WIMG_PREFETCH or self._pipe_active():
            return
        from .model.components.conv_blocks import JLC
        enc = getattr(self.model, "encoder", None)
        ce = getattr(enc, "encoder_conv", None)
        if ce is None:
            return
        ps = self.model.patch_size
        ps = [int(ps)] * 3 if not isinstance(ps, (tuple, list)) else [int(v) for v in ps]
        base = [int(s_) // p_ for s_, p_ in zip(self.x.shape[2:], ps)]
        grid = lambda lvl: [max(g // (2 ** (lvl - 1)), 1) for g in base]
        todo = [(blk, grid(i)) for i in (1, 2, 3, 4) for blk in getattr(ce, f"layer{i}") if isinstance(blk, JLC)]
        decs = [self.model.decoder] + list(getattr(self.model, "rc_decoders", []))
        for lvl in (3, 2, 1):
            for d in decs:
                todo += [(blk, grid(lvl)) for blk in getattr(d, f"layer{lvl}", []) if isinstance(blk, JLC)]
        side = VF.side_stream(self.dev, "weight_images")
        cur = torch.cuda.current_stream(self.dev)
        side.wait_stream(cur)
        self._pf_blocks = [blk for blk, g in todo if VF.jlc_prefetch(blk, g, side)]
        for d in (decs if EXPAND_PREFETCH else []):      # the patch-expand heads (out_conv / out_conv1: 3^3 conv + PixelShuffle): forward and input-gradient images
            head = getattr(d, "out_conv1", None) or getattr(d, "out_conv", None)
            if head is not None:
                VF.expand_prefetch(head[0], side)

    def _prefetch_settle(self):
        """after the encoder-forward stage has joined its side streams: the decoders' blocks need not wait for an event of another stage; entries nobody took are dropped"""
        for blk in getattr(self, "_pf_blocks", []):
            blk.__dict__.pop("_pf_ev", None)

    def _head_bf16(self, k):
        """bf16 storage mode (precision "bf16", functional.BF16_STORAGE): may decoder branch k hand its full-resolution output to the staged loss as a bfloat16 tensor?
        The reconstruction branches always (vx_sqdiff_sum_grad_bs_h takes any shape); the segmentation branch when the fused deep-supervision loss runs its column-owner
        kernels at this geometry (vx_seg_loss_ds_h16_ok), the only ones with 16-bit instances.  Only inside the staged passes: a criterion called on the assembled output
        list (eager single pass, foreign criteria) gets fp32 tensors."""
        if self.precision != "bf16" or not VF.BF16_STORAGE or getattr(self, "_bl", None) is None or VF.get_precision() != "bf16":
            return False
        if k > 0:
            return True
        ok = getattr(self, "_seg_h16_ok", None)
        if ok is None:
            import ctypes
            ok = False
            if getattr(self, "_ds_fused", False) and getattr(self.model.decoder, "deep_supervision", False):
                S = [int(v) for v in self.x.shape[2:]]
                ps = int(self.model.patch_size)
                dims = (ctypes.c_int * 9)(*[max(s // (ps * f), 1) for f in (2, 4, 8) for s in S])
                ok = bool(H.query("vx_seg_loss_ds_h16_ok", ctypes.addressof(dims), 4, int(self.x.shape[0]), self._n_classes(), *S))
            self._seg_h16_ok = ok
        return ok

    def _branch_loss(self):
        """the loss taken apart into per-branch pieces (taped steps with the library's own Loss only; any other criterion runs as one call in _s_loss)"""
        return getattr(self, "_bl", None)

    def _setup_branch_loss(self):
        from .utils.loss import Loss
        self._bl = None
        ok = (self.use_graph and self.replay_mode == "tape" and isinstance(self.criterion, Loss) and hasattr(self.model, "decode_branch")
              and self.model.num_modalities >= 1 and os.environ.get("VELOXSEG_STAGED_LOSS", "1") != "0")
        if ok:
            in_ch = list(self.model.encoder.in_channels)
            self._ch_off = [sum(in_ch[:m]) for m in range(len(in_ch))]
            self._n_heads = None
            self._bl = "pending"            # built at the first enc_fwd, when the number of heads is known from the decoder outputs

    @contextlib.contextmanager
    def _wgrad_side(self):
        """Inside a taped stage the weight-gradient kernels can go to a side stream (csrc/_vxops.cpp WgradSide): in the captured DAG they become a
        branch of their own instead of links of the input-gradient chain.  Off by default (VELOXSEG_TAPE_WGRAD_SIDE): the critical path gets
        0.3 ms shorter and the step slower -- the big weight-gradient grids then take CUs from the chain."""
        m = VF.cpp_module() if (self.use_graph and self.replay_mode == "tape" and TAPE_WGRAD_SIDE) else None
        if m is None:
            yield
            return
        m.set_wgrad_stream(True)
        try:
            yield
        finally:
            m.set_wgrad_stream(False)
            m.wgrad_join(torch.cuda.current_stream(self.dev).cuda_stream, self.dev.index or 0, True)

    def _s_loss(self):
        if self._pipe_active():
            self.flat.grad[self.flat.split:].zero_()       # the decoder half of the flat gradient (see _s_enc_fwd); the decoder forward fan before this stage joined the previous tail
        bl = self._branch_loss()
        if bl is not None:
            n_rc = sum(int(r.numel()) for r in self._rc_c[1:])
            loss = bl.finalize(self._outs[0][-1].detach(), [self._outs[k][1].detach() for k in range(1, len(self._outs))], n_rc)
            self._last_parts = [[o.detach() for o in outs] for outs in self._outs]
            self._last_outputs = None
            torch.add(loss, 0.0, out=self.loss)
            return
        outs_d = [[o.detach().requires_grad_(True) for o in outs] for outs in self._outs]
        self.last_outputs = self.model.assemble_train(outs_d)      # detached leaves: what the per-step metrics read
        loss = self.criterion(self.last_outputs, self.labels, sr_labels=self.x)
        with self._wgrad_side():
            loss.backward()
        self._douts = [[o.grad for o in od] for od in outs_d]
        torch.add(loss.detach(), 0.0, out=self.loss)      # a kernel, not hipMemcpyAsync: memcpy nodes cannot be read back into a launch tape

    def _split_dec_wgrad(self):
        """taped steps: the weight-gradient kernels of a decoder's backward become a tape of their own (dec_wg[k]) that is replayed on the lane the
        encoder backward leaves idle, concurrently with it -- the decoder fans then only carry the input-gradient chains"""
        return self.use_graph and self.replay_mode == "tape" and TAPE_WGRAD_DEFER and VF.cpp_module() is not None

    def _s_dec_bwd(self, k):
        m = VF.cpp_module() if self._split_dec_wgrad() else None
        if m is not None:
            m.set_wgrad_defer(True)              # weight-gradient launches are queued as closures (they own their operands) ...
        try:
            self._s_dec_bwd_body(k)
        finally:
            if m is not None:
                m.set_wgrad_hold()               # ... and stay queued when the stage ends: _s_dec_wg(k) launches them

    def _s_dec_wg(self, k):
        """the queued weight-gradient kernels of decoder k, on the current stream"""
        m = VF.cpp_module()
        m.set_wgrad_defer(True)
        try:
            m.wgrad_join(torch.cuda.current_stream(self.dev).cuda_stream, self.dev.index or 0, True)
        finally:
            m.set_wgrad_defer(False)

    def _s_dec_bwd_body(self, k):
        bl = self._branch_loss()
        if bl is not None:        # this branch's share of the loss backward, then its decoder
            outs = self._outs[k]
            grads = (bl.seg_backward() + [bl.dgs]) if k == 0 else [bl.rc_backward(self._rc_c[k], self.x, self._ch_off[k - 1]), bl.dgm[k - 1]]
            with self._wgrad_side():
                torch.autograd.backward(list(outs), grads)
            return
        pairs = [(o, g) for o, g in zip(self._outs[k], self._douts[k]) if g is not None]
        with self._wgrad_side():
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])

    def _s_enc_bwd(self):
        bt, groups = [], []
        for j, t in enumerate(self._boundary):
            gs = [lv[j].grad for lv in self._leaves if lv[j].grad is not None]
            if gs:
                bt.append(t)
                groups.append(gs)
        bg = self._sum_groups(groups)
        self._stage_stream = torch.cuda.current_stream(self.dev)
        # taped encoder backward: the weight-gradient kernels are deferred to the end of the stream they were queued on (csrc/_vxops.cpp WgradSide,
        # `same`): the mixer / level-input gradients that the OTHER lanes wait for leave the conv chain before its weight gradients run
        m = VF.cpp_module() if (self.use_graph and self.replay_mode == "tape" and TAPE_WGRAD_DEFER) else None
        if m is not None:
            m.set_wgrad_defer(True)
        self._enc_bwd_defer = m is not None
        try:
            with self._wgrad_side():
                torch.autograd.backward(bt, bg)
        finally:
            self._enc_bwd_defer = False
            if m is not None:
                # (VELOXSEG_WGRAD_SPREAD=3 deals these sinks onto the stage's three lanes instead of the stream they were queued from: measured 7.21 vs
                # 6.28 ms per step on one MI355X -- the big weight-gradient grids then land beside the tail of the chain; off)
                m.wgrad_join(self._stage_stream.cuda_stream, self.dev.index or 0, True, int(os.environ.get("VELOXSEG_WGRAD_SPREAD", "0")))
                m.set_wgrad_defer(False)
            self._drop_level_hooks()

    def _sum_groups(self, groups):
        """[g0 (+ g1 (+ g2))] per boundary tensor: the per-branch gradients of one tensor summed; all the 2- and 3-term sums in ONE launch (vx_add_many)"""
        import ctypes
        out = [g[0] if len(g) == 1 else None for g in groups]
        todo = [i for i, g in enumerate(groups) if 2 <= len(g) <= 3 and all(t.is_contiguous() and t.dtype == torch.float32 for t in g)]
        for i, g in enumerate(groups):
            if out[i] is None and i not in todo:
                acc = g[0]
                for h in g[1:]:
                    acc = acc + h
                out[i] = acc
        for lo in range(0, len(todo), 16):
            idx = todo[lo:lo + 16]
            k = len(idx)
            res = [torch.empty_like(groups[i][0]) for i in idx]
            arr = lambda vals: (ctypes.c_void_p * k)(*vals)
            a, b = arr([H.P(groups[i][0]) for i in idx]), arr([H.P(groups[i][1]) for i in idx])
            c = arr([H.P(groups[i][2]) if len(groups[i]) == 3 else None for i in idx])
            o = arr([H.P(r) for r in res])
            n = (ctypes.c_long * k)(*[groups[i][0].numel() for i in idx])
            H.call("vx_add_many", ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c), ctypes.addressof(o), ctypes.addressof(n), k, H.stream_ptr())
            for i, r in zip(idx, res):
                out[i] = r
        return out

    def _eager_stages(self, between=None):
        """the staged pass launched eagerly: decoder stages on forked HIP streams (functional.run_branches; autograd replays each branch's
        backward on the stream of its forward), `between()` is called once the decoder gradients are complete (decoder-bucket all-reduce)"""
        nb = self.model.num_branches
        self._leaves, self._outs = [None] * nb, [None] * nb
        self._s_enc_fwd()
        VF.run_branches([(lambda k=k: self._s_dec_fwd(k)) for k in range(nb)], self.dev, uses=[self._boundary] * nb)
        self._s_loss()
        def dec_bwd(k):
            self._s_dec_bwd(k)
            if self._split_dec_wgrad():
                self._s_dec_wg(k)
        VF.run_branches([(lambda k=k: dec_bwd(k)) for k in range(nb)], self.dev, uses=[([] if self._branch_loss() is not None else self._douts[k]) for k in range(nb)])
        if between is not None:
            between()
        self._s_enc_bwd()

    def _eager_pass(self):
        self._eager_stages()

    def _check_tiling(self):
        """the slices reduced in one step must tile [0, numel) exactly once (ranks that disagree on this hang or diverge)"""
        got = sorted(self._reduced or [])
        pos = 0
        for lo, hi in got:
            if lo != pos:
                break
            pos = hi
        if pos != self.flat.numel or not got:
            raise RuntimeError(f"TrainEngine: the all-reduce slices of this step {got} do not tile the flat gradient buffer [0, {self.flat.numel})")

    def _allreduce(self, lo, hi):
        if getattr(self, "_reduced", None) is not None:
            self._reduced.append((lo, hi))            # (tests: the slices of one step must tile the flat buffer exactly once)
        if self.skip_comm:                            # A/B leg of the comm diagnostics (bench.py `comm.step_ms_no_comm`): every rank skips the same collectives
            return
        prof = self.comm_profile
        if prof is not None:
            # HIP events on the stream the collective is issued from: with RCCL the work runs on the process group's own stream, which first waits for this
            # stream and which this stream waits for afterwards (synchronous op) -- so e0 -> e1 on THIS stream brackets the bucket from "its gradients are
            # complete" to "reduced values visible", queueing on the hardware queue it landed on included
            st = torch.cuda.current_stream(self.dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
        dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
        if prof is not None:
            e1.record(st)
            prof.append({"lo": int(lo), "hi": int(hi), "bytes": int(hi - lo) * 4, "stream": int(st.cuda_stream), "e0": e0, "e1": e1})

    def comm_report(self):
        """Summary of the collectives recorded since `comm_profile = []` was set: per bucket (lo, hi) the bytes, the mean / max enqueue -> complete time on
        the issuing stream, which stream that was (the dec_wg lane, the engine's comm stream or the caller's) and how many were recorded.  Synchronises."""
        prof = self.comm_profile or []
        torch.cuda.synchronize(self.dev)
        lanes = {}
        try:
            for i, s_ in enumerate(self._lane_streams(4)):
                lanes[int(s_.cuda_stream)] = f"tape lane {i}"
        except Exception:
            pass
        if self.comm_stream is not None:
            lanes[int(self.comm_stream.cuda_stream)] = "engine comm stream"
        by = {}
        for r in prof:
            d = by.setdefault((r["lo"], r["hi"]), {"lo": r["lo"], "hi": r["hi"], "bytes": r["bytes"], "ms": [], "stream": lanes.get(r["stream"], "caller stream")})
            d["ms"].append(r["e0"].elapsed_time(r["e1"]))
        out = []
        for d in by.values():
            ms = d.pop("ms")
            d.update({"count": len(ms), "ms_mean": round(sum(ms) / len(ms), 4), "ms_max": round(max(ms), 4)})
            out.append(d)
        return sorted(out, key=lambda d: -d["lo"])

    # ---- torch.optim.AdamW as the state carrier (reference checkpoints, LR schedulers) ---------------
    def bind_optimizer(self, optimizer):
        """`optimizer` = torch.optim.AdamW over model.parameters() (utils/optimizers/optimizers.py:27-34).  Its param_groups[0] becomes the
        source of lr / betas / eps / weight_decay (so the reference's per-epoch LR schedulers drive the fused HIP update), and its
        per-parameter state is made to alias the engine's flat moment buffers, so optimizer.state_dict() / load_state_dict() read and
        write the reference's checkpoint format (utils/load_model.py:90-109).  Call again (or load_optimizer_state) after
        optimizer.load_state_dict()."""
        if not isinstance(optimizer, torch.optim.AdamW):
            raise TypeError("TrainEngine drives torch.optim.AdamW only (the optimizer of every shipped train_config)")
        if len(optimizer.param_groups) != 1 or sorted(id(p) for p in optimizer.param_groups[0]["params"]) != sorted(id(p) for p in self.flat.params):
            raise ValueError("optimizer must hold exactly model.parameters() in one param group")
        if optimizer.param_groups[0].get("amsgrad") or optimizer.param_groups[0].get("maximize"):
            raise NotImplementedError("amsgrad / maximize are not supported by the fused AdamW")
        self.optimizer = optimizer
        self.load_optimizer_state()

    def load_optimizer_state(self):
        """copy whatever state the bound optimizer holds (e.g. after load_state_dict) into the flat buffers, then alias it back"""
        opt = self.optimizer
        steps = []
        for n, p in zip(self.flat.names, self.flat.params):
            o, k = self.flat.slices[n]
            st = opt.state.get(p, {})
            if "exp_avg" in st:
                self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(int(float(st["step"])))
        if steps:
            if len(set(steps)) != 1 or len(steps) != len(self.flat.params):
                raise ValueError("optimizer state has differing per-parameter step counts; the fused AdamW keeps one")
            self.t = steps[0]
        self._step_tensor = torch.tensor(float(self.t), dtype=torch.float32)
        for n, p in zip(self.flat.names, self.flat.params):
            o, k = self.flat.slices[n]
            opt.state[p] = {"step": self._step_tensor, "exp_avg": self.m[o:o + k].view(p.shape), "exp_avg_sq": self.v[o:o + k].view(p.shape)}

    def _adamw_begin(self):
        """once per step: hyper-parameters from the bound optimizer, step count"""
        if self.optimizer is not None:
            g = self.optimizer.param_groups[0]
            self.lr, self.wd, self.betas, self.eps = g["lr"], g["weight_decay"], g["betas"], g["eps"]
            self._step_tensor += 1
            self.optimizer._opt_called = True        # the fused kernel below IS optimizer.step(); keeps LR schedulers from warning
        self.t += 1

    def _adamw_range(self, lo, hi):
        """the fused update of flat[lo:hi] on the current stream (element-wise: any slice of the flat buffers)"""
        if hi <= lo:
            return
        H.call("vx_adamw_step", H.P(self.flat.param[lo:hi]), H.P(self.flat.grad[lo:hi]), H.P(self.m[lo:hi]), H.P(self.v[lo:hi]), int(hi - lo), float(self.lr),
               float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), self.t, 1.0 / self.world, H.stream_ptr())

    def _adamw(self):
        self._adamw_begin()
        self._adamw_range(0, self.flat.numel)

    def _check_flag_timeouts(self):
        if self.replay_mode == "tape":
            nto = H.query("vx_tape_flag_timeouts")         # a poll of an EARLIER replay that gave up (pinned host word): never apply an update on top of it
            if nto:
                raise RuntimeError(f"TrainEngine.step: {nto} cross-lane poll(s) gave up after the flag timeout; the gradients of that step are not trustworthy "
                                   "and no optimizer update was applied for this one (VELOXSEG_TAPE_FLAG_TIMEOUT_MS / vx_tape_set_flag_timeout_ms)")

    def flush(self):
        """pipelined tail: order the CURRENT stream behind the decoder weight gradients / all-reduce / AdamW half still running on the fourth lane (call before reading
        decoder parameters or gradients on this stream: checkpoints, validation forward, EMA copies)"""
        if self._tail_pending and self.graphs is not None and "dec_wg" in self.graphs:
            self._hop(43, self._lane_streams(4)[3], torch.cuda.current_stream(self.dev))

    # ---- capture --------------------------------------------------------------------------------
    def _forked(self, fn, *args):
        """Run `fn` with a trivial side-stream branch alive around it, so that every captured graph has more than one branch.
        ROCm 7.2 replays single-branch graphs through a batched AQL-packet path that intermittently corrupts this step after a
        device synchronise (veloxseg_amd/__init__.py); multi-branch graphs take the ordinary per-node path."""
        if os.environ.get("VX_NO_FORK") == "1":         # debugging only (tools/graph_replay_repro.py)
            return fn(*args)
        cur = torch.cuda.current_stream(self.dev)
        self._fork_stream.wait_stream(cur)
        with torch.cuda.stream(self._fork_stream):
            self._fork_buf.add_(1.0)
        fn(*args)
        cur.wait_stream(self._fork_stream)

    def _graph(self, pool, fn, *args, lanes=None):
        tape = self.replay_mode == "tape"
        g = torch.cuda.CUDAGraph(keep_graph=True) if tape else torch.cuda.CUDAGraph()
        # with RCCL running (world > 1) its watchdog thread polls events while we capture: in the default "global" error mode that invalidates the
        # capture; "thread_local" checks only the capturing thread (kernels queued by autograd's device thread are captured either way)
        mode = "thread_local" if (self.dp or os.environ.get("VELOXSEG_CAPTURE_MODE") == "thread_local") else "global"
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                if tape:
                    fn(*args)                      # the tape launches node by node: no need for the second branch that keeps hipGraphLaunch correct
                else:
                    self._forked(fn, *args)
        for w_ in caught:                          # a stage may hold no launches in some configurations (e.g. dec_wg[k] of a decoder without deferred
            if "Graph is empty" not in str(w_.message):      # weight gradients): that is an _EmptyTape below, not a warning for the user
                warnings.warn_explicit(w_.message, w_.category, w_.filename, w_.lineno)
        if not tape:
            return g
        t = LaunchTape(g, int(lanes or self.tape_lanes))
        return t if t.n_nodes > 0 else _EmptyTape()

    def _capture(self):
        """One hipGraph per stage.  Branch graphs allocate from their own memory pool (a shared pool hands the blocks one capture freed
        to the next capture, which is only safe when the graphs replay in capture order, not concurrently); tensors that cross
        stages stay referenced by the engine for the lifetime of the graphs."""
        self.model.train()
        rng = VF.rng_state(self.dev)
        rng0 = rng.clone()
        self.flat.reattach()
        nb = self.model.num_branches
        self._fork_stream = torch.cuda.Stream(device=self.dev)
        self._fork_buf = torch.zeros(64, device=self.dev)
        self.branch_streams = [torch.cuda.Stream(device=self.dev) for _ in range(nb)]
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(1, self._warm)):   # warm allocator / lazy inits outside the capture; the last pass is the reference
                rng.copy_(rng0)
                self._eager_pass()
            ref = (self.loss.double().clone(), self.flat.grad.double().abs().sum())
        torch.cuda.current_stream().wait_stream(s)
        rng.copy_(rng0)                           # the warm-up steps must not consume dropout streams: graph == eager run
        torch.cuda.synchronize()
        self._leaves, self._outs = [None] * nb, [None] * nb
        main_pool = torch.cuda.graph_pool_handle()
        pools = [torch.cuda.graph_pool_handle() for _ in range(nb)]
        G = {"enc_fwd": self._graph(main_pool, self._s_enc_fwd)}
        G["dec_fwd"] = [self._graph(pools[k], self._s_dec_fwd, k) for k in range(nb)]
        G["loss"] = self._graph(main_pool, self._s_loss)
        if self._split_dec_wgrad():
            G["dec_bwd"], G["dec_wg"] = [], []
            for k in range(nb):                   # (pairwise: the closures queued by dec_bwd[k] are launched by dec_wg[k])
                G["dec_bwd"].append(self._graph(pools[k], self._s_dec_bwd, k))
                G["dec_wg"].append(self._graph(pools[k], self._s_dec_wg, k))
        else:
            G["dec_bwd"] = [self._graph(pools[k], self._s_dec_bwd, k) for k in range(nb)]
        # the encoder backward shares the GPU with the dec_wg tapes, which are replayed on the 4th lane stream: keep it on the other three (a tape laid
        # out on four lanes would put one of its chains behind ~2 ms of decoder weight gradients)
        G["enc_bwd"] = self._graph(main_pool, self._s_enc_bwd, lanes=3 if "dec_wg" in G else None)
        self.graphs = G
        if self.replay_mode == "tape" and TAPE_PGO:
            # profile-guided layout of the multi-lane tapes: one replay fills every buffer with real values, then each node of the encoder tapes is timed
            # alone and the tape is laid out again with those durations; the self-check below runs on the new layout
            rng.copy_(rng0)
            self._replay(comm=False)
            torch.cuda.synchronize()
            for key in TAPE_PGO_STAGES:
                t = G.get(key)
                if isinstance(t, LaunchTape):
                    t.relayout(3 if (key == "enc_bwd" and "dec_wg" in G) else self.tape_lanes)
            torch.cuda.synchronize()
        # self-check: replays separated by device synchronisation must reproduce the eager pass (same dropout streams)
        for k in range(self.verify_replays):
            rng.copy_(rng0)
            torch.cuda.synchronize()
            self._replay(comm=False)
            torch.cuda.synchronize()
            got = (self.loss.double(), self.flat.grad.double().abs().sum())
            bad = [not bool(torch.isfinite(a)) or abs(float(a) - float(b)) > 2e-3 * max(abs(float(b)), 1e-30) for a, b in zip(got, ref)]
            if any(bad):
                warnings.warn(f"hipGraph replay {k} does not reproduce the eager step (loss {float(got[0]):.6g} vs {float(ref[0]):.6g}, "
                              f"|grad| {float(got[1]):.6g} vs {float(ref[1]):.6g}); TrainEngine falls back to eager launches")
                self.graphs, self.use_graph = None, False
                break
        rng.copy_(rng0)

    def _lane_streams(self, n):
        """streams for n concurrent tapes: the tape's process-wide lane streams, which sit on different hardware queues (csrc/tape.hip: ROCm
        multiplexes all streams onto 4 hardware queues, streams that share one never overlap, and nothing overlaps with the NULL stream)"""
        import ctypes
        cache = self.__dict__.setdefault("_lane_cache", {})
        if n not in cache:
            out = []
            for k in range(n):
                h = ctypes.c_void_p()
                H.call("vx_tape_lane_stream", H.stream_ptr(), k, ctypes.addressof(h))
                out.append(torch.cuda.ExternalStream(h.value, device=self.dev))
            cache[n] = out
        return cache[n]

    def _hop(self, slot, src, dst):
        """dst waits for what is enqueued on src: in tape mode through the tape's flag kernels (csrc/tape.hip vx_tape_hop: ~2 us of queue time per hop
        against ~14 us for an event record + wait on this runtime), else an event"""
        if self.replay_mode == "tape":
            H.call("vx_tape_hop", int(slot), src.cuda_stream, dst.cuda_stream)
        else:
            dst.wait_stream(src)

    def wg_order(self):
        """the order in which the dec_wg tapes are queued on the fourth lane: decoders with the shorter backward tape first (they finish first)"""
        G = self.graphs
        o = getattr(self, "_wg_order", None)
        if o is None:
            o = self._wg_order = sorted(range(len(G["dec_bwd"])), key=lambda k: (G["dec_bwd"][k].n_nodes, k))
        return o

    def _fan(self, graphs, slot0=0):
        cur = torch.cuda.current_stream(self.dev)
        streams = self._lane_streams(len(graphs)) if self.replay_mode == "tape" else self.branch_streams
        for k, (s_, g) in enumerate(zip(streams, graphs)):
            self._hop(slot0 + k, cur, s_)
            with torch.cuda.stream(s_):
                g.replay()
        for k, s_ in enumerate(streams[:len(graphs)]):
            self._hop(slot0 + 8 + k, s_, cur)

    def _replay(self, comm: bool, adamw: bool = False):
        """enc_fwd, {dec_fwd}, loss, {dec_bwd}, {dec_wg on the fourth lane || enc_bwd}, then -- enqueued AFTER the encoder-backward tape -- the decoder bucket's
        all-reduce on the dec_wg lane's stream (it runs behind the weight gradients, beside the rest of the encoder backward) and the encoder bucket's"""
        G = self.graphs
        cur = torch.cuda.current_stream(self.dev)
        G["enc_fwd"].replay()
        pipe = adamw and self._pipe_active() and "dec_wg" in G
        if self._tail_pending:
            # the previous step's tail (dec_wg tapes, decoder all-reduce, decoder AdamW on the fourth lane) must be done before the decoders run again
            self._hop(42, self._lane_streams(4)[3], cur)
            self._tail_pending = False
        self._fan(G["dec_fwd"], 0)
        G["loss"].replay()
        split, n = self.flat.split, self.flat.numel
        wg_lane = None
        if "dec_wg" in G and self.replay_mode == "tape" and WG_EARLY and len(G["dec_bwd"]) <= 3:
            # The decoders' weight gradients run one after the other on the fourth lane, beside the encoder backward on the other three.  dec_wg[k] needs dec_bwd[k]
            # only, and the fourth lane is idle during the decoder-backward fan: each dec_wg[k] is queued behind ITS decoder's backward (a hop from that lane), the
            # decoders that finish first (the reconstruction branches: shorter tapes) first -- the weight gradients start while the longest decoder is still in its
            # backward instead of after the join of all three (the phase "dec_wg beside enc_bwd" was the longest of the step: 1.89 ms of 4.2).
            streams = self._lane_streams(len(G["dec_bwd"]))
            for k, (s_, g) in enumerate(zip(streams, G["dec_bwd"])):
                self._hop(16 + k, cur, s_)
                with torch.cuda.stream(s_):
                    g.replay()
            wg_lane = self._lane_streams(4)[3]
            for k in self.wg_order():
                self._hop(44 + k, streams[k], wg_lane)
                with torch.cuda.stream(wg_lane):
                    G["dec_wg"][k].replay()
            for k, s_ in enumerate(streams):
                self._hop(24 + k, s_, cur)
        else:
            self._fan(G["dec_bwd"], 16)
            if "dec_wg" in G:
                # the decoders' weight gradients: one after the other on the lane the encoder backward does not use, while it runs on the other three
                wg_lane = self._lane_streams(4)[3]
                self._hop(40, cur, wg_lane)
                with torch.cuda.stream(wg_lane):
                    for t in G["dec_wg"]:
                        t.replay()
        if comm:
            self._reduced = []
        early = comm and self.dp and self.overlap and self.comm_placement == "fresh_before" and wg_lane is not None
        if early:                                               # (diagnostic placement: see comm_placement)
            self.comm_stream.wait_stream(wg_lane)
            with torch.cuda.stream(self.comm_stream):
                self._allreduce(split, n)
        G["enc_bwd"].replay()
        if comm and self.dp:
            if self.overlap:
                # WHERE the collectives are enqueued decides whether they cost their own duration or a millisecond (tools/comm_standin_probe.py, one MI355X,
                # a 150 us stand-in kernel for the decoder bucket): ROCm multiplexes every stream onto 4 hardware queues that run their packets in order, so a
                # stream that WAITS for a long dependency blocks the queue it shares from the moment the wait is enqueued.  The decoder bucket is complete
                # only when the dec_wg tapes (the decoders' weight gradients, ~1 ms on the fourth lane) are done; enqueued BEFORE the encoder-backward tape
                # on a stream of its own, that wait stalled one of the tape's lanes: +1.07 ms per step.  Enqueued AFTER the tape, on the dec_wg lane's own
                # stream (stream order instead of an event wait): +0.09 ms -- it runs behind the weight gradients while the encoder backward still has
                # ~1 ms to go.  More hardware queues are no way out (GPU_MAX_HW_QUEUES 5 / 6: the step itself 6.25 -> 6.95 / 8.2 ms).
                cs = wg_lane if (wg_lane is not None and self.comm_placement == "lane") else self.comm_stream
                if wg_lane is None:
                    cs.wait_stream(cur)
                elif cs is not wg_lane:
                    cs.wait_stream(wg_lane)
                if not early:
                    with torch.cuda.stream(cs):
                        self._allreduce(split, n)               # decoder bucket: behind the dec_wg tapes, beside the rest of the encoder backward
                tape = G["enc_bwd"] if self.replay_mode == "tape" else None
                markers = [t for t in (3, 2, 1) if tape is not None and H.query("vx_tape_has_marker", tape.handle, int(t))]
                for trig, lo, hi in self.flat.taped_schedule(self.bucket_min_bytes, markers)[1:]:      # ([0] = the decoder bucket, enqueued above)
                    if trig != 0:
                        # (opt-in per-level buckets) the tape recorded an event where this level's gradients were complete
                        H.call("vx_tape_wait_marker", tape.handle, int(trig), cs.cuda_stream)
                    else:
                        if pipe:
                            cs = self.comm_stream           # (pipelined tail: the encoder bucket must not sit behind the fourth lane's weight gradients)
                        cs.wait_stream(cur)                 # the encoder's gradients (one bucket by default): after the tape
                    with torch.cuda.stream(cs):
                        self._allreduce(lo, hi)
                if pipe:
                    cur.wait_stream(self.comm_stream)
                else:
                    cur.wait_stream(cs)
                self._check_tiling()
            else:
                if wg_lane is not None:
                    cur.wait_stream(wg_lane)
                self._allreduce(0, n)
        if pipe and not (comm and self.dp and not self.overlap):
            # pipelined tail: decoder AdamW on the fourth lane behind dec_wg (and the decoder bucket's all-reduce), encoder AdamW on the caller's stream; the lane is
            # joined before the NEXT decoder forward (hop 42 above) or by flush()
            self._check_flag_timeouts()
            self._adamw_begin()
            if comm and self.dp and self.overlap and self.comm_placement != "lane":
                wg_lane.wait_stream(self.comm_stream)          # (diagnostic placements: the decoder bucket was reduced on the comm stream)
            with torch.cuda.stream(wg_lane):
                self._adamw_range(split, n)
            self._adamw_range(0, split)
            self._tail_pending = True
            return
        if wg_lane is not None:
            self._hop(41, wg_lane, cur)
        if adamw:
            self._check_flag_timeouts()
            self._adamw()

    def _check_labels(self, labels):
        """The engine's label volume is uint8 when the class count allows (one byte per voxel instead of the loader's int64): a narrowing copy would WRAP an
        out-of-range value (a 255 / -1 ignore label, BraTS label 4 left unmapped) into a valid class id, where the reference's CrossEntropyLoss raises a device assert
        (utils/loss.py:17).  The first label batches (VELOXSEG_LABEL_CHECK: "first" = 4 batches, the default; "always"; "off") are range-checked before the copy -- one
        small reduction + a host read each; VELOXSEG_LABELS=int64 keeps the loader's dtype instead."""
        mode = os.environ.get("VELOXSEG_LABEL_CHECK", "first")
        n = getattr(self, "_label_checks", 0)
        if mode == "off" or (mode != "always" and n >= 4) or labels.dtype == self.labels.dtype:
            return
        self._label_checks = n + 1
        ncls = self._n_classes() if hasattr(self.model, "n_classes") or hasattr(self.model, "decoder") else None
        lo, hi = int(labels.min()), int(labels.max())
        top = (ncls - 1) if ncls else torch.iinfo(self.labels.dtype).max
        if lo < 0 or hi > top:
            raise ValueError(f"TrainEngine.step: label values span [{lo}, {hi}] but the model has classes 0..{top}; the engine's {self.labels.dtype} label buffer would wrap them "
                             "silently (map ignore / raw labels first, e.g. BraTS 4 -> 3, or run with VELOXSEG_LABELS=int64)")

    # ---- public ---------------------------------------------------------------------------------
    def step(self, x: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimisation step.  x / labels are copied into the engine's static buffers (None = reuse their contents)."""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if labels is not None:
            self._check_labels(labels)
            self.labels.copy_(labels, non_blocking=True)
        if not self.model.training:
            self.model.train()
        VF.weights_epoch_bump()                                # (weight images kept by a TapedPredictor of this model are stale after this step: its AdamW writes through raw pointers)
        if self.use_graph and self.graphs is None:
            try:
                with self._settings(capture=True):
                    self._capture()                            # may clear use_graph (self-check)
            except Exception as e:                             # a capture that cannot be taken (or read back) must not take the training run down
                warnings.warn(f"TrainEngine: capturing the step failed ({type(e).__name__}: {str(e)[:300]}); falling back to eager launches")
                self.graphs, self.use_graph = None, False
                torch.cuda.synchronize()
        if self.use_graph:
            self._replay(comm=True, adamw=True)                # (everything the tapes need is baked in: no process-wide state is read); the optimizer update included
            return self.loss
        with self._settings():
            cur = torch.cuda.current_stream(self.dev)
            self.flat.reattach()
            split, n = self.flat.split, self.flat.numel
            if not self.dp:
                self._fwd_bwd_single()
            elif not self.overlap:
                self._fwd_bwd_single()
                self._allreduce(0, n)
            else:
                self._fwd_bwd_overlapped()                      # every bucket but the last is reduced while the backward pass still runs
        self._adamw()
        return self.loss


def ddp_average_gradients(params, world_size, group=None):
    """Plain helper (used by the gloo CPU tests): in-place average of .grad over ranks via one flat all-reduce."""
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world_size
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
