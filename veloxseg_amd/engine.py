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

import warnings
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

import veloxseg_amd as _pkg
from . import _hip as H
from . import functional as VF


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class FlatParams:
    """Re-homes every parameter (and its .grad) of `model` as a view into one flat fp32 buffer."""

    def __init__(self, model: torch.nn.Module, first: Sequence[str] = ("encoder.",)):
        params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0][1].device
        head = [(n, p) for n, p in params if any(n.startswith(f) for f in first)]
        tail = [(n, p) for n, p in params if not any(n.startswith(f) for f in first)]
        self.names: List[str] = []
        self.slices = {}
        off = 0
        for n, p in head + tail:
            self.slices[n] = (off, p.numel())
            self.names.append(n)
            off = _align(off + p.numel())
        self.split = self.slices[tail[0][0]][0] if tail else off
        self.numel = off
        self.param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for n, p in head + tail:
                o, k = self.slices[n]
                self.param[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.param[o:o + k].view(p.shape)
                p.grad = self.grad[o:o + k].view(p.shape)
        self.params = [p for _, p in head + tail]

    def zero_grad(self):
        self.grad.zero_()

    def reattach(self):
        """make sure .grad still aliases the flat buffer (e.g. after zero_grad(set_to_none=True))"""
        for n, p in zip(self.names, self.params):
            o, k = self.slices[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + k].view(p.shape)


class TrainEngine:
    """step(x, labels) = zero_grad -> forward -> loss -> backward -> [all-reduce] -> AdamW, on static buffers."""

    def __init__(self, model, criterion, batch_shape, label_dtype=torch.int64, lr=2.5e-4, weight_decay=0.01, betas=(0.9, 0.999),
                 eps=1e-8, use_graph=True, overlap=True, process_group=None, warmup_steps=2):
        self.model, self.criterion = model, criterion
        self.dev = next(model.parameters()).device
        self.flat = FlatParams(model)
        self.m = torch.zeros_like(self.flat.param)
        self.v = torch.zeros_like(self.flat.param)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.overlap = overlap and self.world > 1
        if use_graph and not _pkg.GRAPH_REPLAY_SAFE:
            warnings.warn("veloxseg_amd was imported after the HIP runtime initialised without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; "
                          "hipGraph replay is unsafe on this ROCm (see veloxseg_amd/__init__.py) -- TrainEngine launches eagerly")
            use_graph = False
        self.use_graph = use_graph
        B = batch_shape[0]
        self.x = torch.zeros(batch_shape, device=self.dev, dtype=torch.float32)
        self.labels = torch.zeros((B, 1, *batch_shape[2:]), device=self.dev, dtype=label_dtype)
        self.loss = torch.zeros((), device=self.dev, dtype=torch.float32)
        self.graphs = None
        self.comm_stream = torch.cuda.Stream(device=self.dev) if self.world > 1 else None
        self._warm = warmup_steps
        if self.world > 1:
            dist.broadcast(self.flat.param, src=0, group=self.pg)      # identical replicas at start

    # ---- pieces ---------------------------------------------------------------------------------
    def _forward_loss(self):
        outs = self.model(self.x)
        return outs, self.criterion(outs, self.labels, sr_labels=self.x)

    def _fwd_bwd_single(self):
        self.flat.zero_grad()
        _, loss = self._forward_loss()
        loss.backward()
        self.loss.copy_(loss.detach())

    def _phase1(self):
        """forward + loss + backward through the decoders down to the encoder outputs"""
        self.flat.zero_grad()
        VF.advance_rng(self.dev)
        enc = self.model.encoder
        attn, encs = enc(self.x)
        # cut the autograd graph at the encoder outputs: the decoders consume detached leaves, so this phase touches decoder nodes
        # only; the leaves' .grad then seed the encoder backward (phase 2), which adds the encoder-internal paths (enc_i -> down_{i+1}).
        encs_d = [e.detach().requires_grad_(True) for e in encs]
        attn_d = [[t.detach().requires_grad_(True) for t in lvl] for lvl in attn]
        self._boundary = list(encs) + [t for lvl in attn for t in lvl]
        leaves = encs_d + [t for lvl in attn_d for t in lvl]
        outs = self._decode(attn_d, encs_d)
        loss = self.criterion(outs, self.labels, sr_labels=self.x)
        loss.backward()
        self._bgrads = [l.grad for l in leaves]
        self.loss.copy_(loss.detach())

    def _decode(self, attn, encs):
        m = self.model
        pred, dec_pram = m.decoder(*encs)
        pred = [m.scale_prediction(p) for p in pred]
        rcs, prams = [], []
        for k in range(m.num_modalities):
            rc, pr = m.rc_decoders[k]([attn[L][k] for L in range(4)], encs)
            rcs.append(rc)
            prams.append(pr)
        rcs = rcs[0] if len(rcs) == 1 else torch.cat(rcs, dim=1)
        return pred + [rcs] + [dec_pram] + prams

    def _phase2(self):
        torch.autograd.backward(self._boundary, self._bgrads)
        self._boundary = self._bgrads = None

    def _allreduce(self, lo, hi):
        dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)

    def _adamw(self):
        self.t += 1
        H.call("vx_adamw_step", H.P(self.flat.param), H.P(self.flat.grad), H.P(self.m), H.P(self.v), self.flat.numel, float(self.lr),
               float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), self.t, 1.0 / self.world, H.stream_ptr())

    # ---- capture --------------------------------------------------------------------------------
    def _capture(self):
        self.model.train()
        rng0 = VF.rng_state(self.dev).clone()
        self.flat.reattach()
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warm):           # warm allocator / lazy inits outside the capture
                if self.overlap:
                    self._phase1()
                    self._phase2()
                else:
                    self._fwd_bwd_single()
        torch.cuda.current_stream().wait_stream(s)
        VF.rng_state(self.dev).copy_(rng0)        # the warm-up steps must not consume dropout streams: graph == eager run
        torch.cuda.synchronize()
        if self.overlap:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._phase1()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._phase2()
            self.graphs = (g1, g2)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._fwd_bwd_single()
            self.graphs = (g,)

    # ---- public ---------------------------------------------------------------------------------
    def step(self, x: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimisation step.  x / labels are copied into the engine's static buffers (None = reuse their contents)."""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        self.model.train()
        if self.use_graph and self.graphs is None:
            self._capture()
        cur = torch.cuda.current_stream()
        if self.world == 1:
            if self.use_graph:
                self.graphs[0].replay()
            else:
                self.flat.reattach()
                self._fwd_bwd_single()
        elif not self.overlap:
            if self.use_graph:
                self.graphs[0].replay()
            else:
                self.flat.reattach()
                self._fwd_bwd_single()
            self._allreduce(0, self.flat.numel)
        else:
            split, n = self.flat.split, self.flat.numel
            if self.use_graph:
                self.graphs[0].replay()
            else:
                self.flat.reattach()
                self._phase1()
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                self._allreduce(split, n)                   # decoder bucket, overlapped with the encoder backward
            if self.use_graph:
                self.graphs[1].replay()
            else:
                self._phase2()
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                self._allreduce(0, split)                   # encoder bucket
            cur.wait_stream(self.comm_stream)
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
