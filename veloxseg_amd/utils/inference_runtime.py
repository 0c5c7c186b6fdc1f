"""Sliding-window inference on MI355X -- drop-in for the reference's utils/inference_runtime.py.

`sliding_window_predict(inputs, predictor, roi_size, sw_batch_size, test_config, **kwargs)` keeps the reference signature
(utils/inference_runtime.py:4-19); the MONAI call underneath it (monai.inferers.sliding_window_inference, constant blending,
constant padding) is re-implemented natively: window origins on the host, window extraction / blending / normalisation / argmax in
HIP kernels (csrc/infer.hip), the predictor being the HIP VeloxSeg forward.  Semantics restated in oracle/sliding_window_oracle.py.
"""
from __future__ import annotations

import math
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .. import _hip as H


def _valid_patch(image_size, patch_size):
    return tuple(min(m, p) if p else m for m, p in zip(image_size, patch_size))


def scan_interval(image_size, roi_size, overlap: float) -> Tuple[int, ...]:
    """int(roi * (1 - overlap)) >= 1 per axis; the whole roi where it spans the axis (monai `_get_scan_interval`)"""
    out = []
    for m, r in zip(image_size, roi_size):
        if r == m:
            out.append(int(r))
        else:
            iv = int(r * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def window_starts(image_size, roi_size, interval) -> List[Tuple[int, int, int]]:
    """window origins, first axis slowest, last window of every axis clamped to the edge (monai `dense_patch_slices`)"""
    per_axis = []
    for m, r, iv in zip(image_size, roi_size, interval):
        if iv == 0:
            n = 1
        else:
            num = int(math.ceil(float(m) / iv))
            first = next((d for d in range(num) if d * iv + r >= m), None)
            n = first + 1 if first is not None else 1
        per_axis.append([k * iv - max(k * iv + r - m, 0) for k in range(n)])
    return [(a, b, c) for a in per_axis[0] for b in per_axis[1] for c in per_axis[2]]


def _axis_counts(starts_axis: Sequence[int], r: int, m: int, device) -> torch.Tensor:
    c = torch.zeros(m, dtype=torch.float32)
    for s in starts_axis:
        c[s:s + r] += 1.0
    return c.to(device)


# two window batches in flight (round 6): each forward of a window batch is a ~250-launch tape that leaves most of the chip idle at the reference's sw_batch_size of 2; the next
# batch's forward is replayed from a second caller stream on a replica of the tape (engine.TapedPredictor.forward_slot) while the first one runs.  The blending stays in the
# reference's window order -- the accumulation of batch i + 1 waits for the one of batch i -- so the result is bit for bit the sequential one.  VELOXSEG_SW_PIPELINE=0: sequential.
SW_PIPELINE = os.environ.get("VELOXSEG_SW_PIPELINE", "1") != "0"
SW_INFLIGHT = max(2, min(4, int(os.environ.get("VELOXSEG_SW_INFLIGHT", "2"))))      # window batches in flight (A/B)


def _pipelined(inputs, predictor, starts, num_win, total, swb, cdhw, roi):
    C, D, Hh, W = cdhw
    rd, rh, rw = roi
    dev = inputs.device
    cur = torch.cuda.current_stream(dev)
    streams = _SW_STREAMS.setdefault(str(dev), [torch.cuda.Stream(device=dev) for _ in range(SW_INFLIGHT)])
    wins = [torch.empty((swb, C, rd, rh, rw), device=dev, dtype=torch.float32) for _ in range(SW_INFLIGHT)]
    # the first batch the plain way (captures / validates the tape of this shape; tells us K)
    g0 = 0
    for k in range(swb):
        z0, y0, x0 = starts[k % num_win]
        H.call("vx_sw_extract", H.P(inputs[k // num_win]), H.P(wins[0][k]), C, D, Hh, W, rd, rh, rw, z0, y0, x0, H.stream_ptr())
    prob = predictor(wins[0])
    if isinstance(prob, (list, tuple)):
        prob = prob[0]
    if prob.dtype != torch.float32 or not prob.is_contiguous() or tuple(prob.shape[2:]) != (rd, rh, rw):
        return None
    K = int(prob.shape[1])
    B = inputs.shape[0]
    acc = torch.zeros((B, K, D, Hh, W), device=dev, dtype=torch.float32)
    for k in range(swb):
        z0, y0, x0 = starts[k % num_win]
        H.call("vx_sw_accumulate", H.P(prob[k]), H.P(acc[k // num_win]), K, D, Hh, W, rd, rh, rw, z0, y0, x0, 1.0, H.stream_ptr())
    # the rest alternately on the two slot streams
    for s_ in streams:
        s_.wait_stream(cur)
    last_acc = None                                          # event after the accumulation of the previous batch
    nb = total // swb
    for bi in range(1, nb):
        # replicas 1 and 2 of the tape, each always replayed from ITS stream (replica 0 -- the plain call above -- stays on the caller's stream: a tape whose caller stream
        # changes between replays synchronises the host once)
        slot = bi % SW_INFLIGHT
        st_ = streams[slot]
        with torch.cuda.stream(st_):
            for k in range(swb):
                i = bi * swb + k
                z0, y0, x0 = starts[i % num_win]
                H.call("vx_sw_extract", H.P(inputs[i // num_win]), H.P(wins[slot][k]), C, D, Hh, W, rd, rh, rw, z0, y0, x0, st_.cuda_stream)
            out = predictor.forward_slot(wins[slot], 1 + slot)
            if out is None:                                  # no replica (capture refused): this batch on the eager launches, on this stream
                out = predictor.model(wins[slot]) if hasattr(predictor, "model") else predictor(wins[slot])
            if isinstance(out, (list, tuple)):
                out = out[0]
            out = out.contiguous().float()
            if last_acc is not None:
                st_.wait_event(last_acc)                     # blending in window order: bit for bit the sequential sums
            for k in range(swb):
                i = bi * swb + k
                z0, y0, x0 = starts[i % num_win]
                H.call("vx_sw_accumulate", H.P(out[k]), H.P(acc[i // num_win]), K, D, Hh, W, rd, rh, rw, z0, y0, x0, 1.0, st_.cuda_stream)
            last_acc = torch.cuda.Event()
            last_acc.record(st_)
    for s_ in streams:
        cur.wait_stream(s_)
    return acc, K


_SW_STREAMS = {}


def sliding_window_inference(inputs: torch.Tensor, roi_size: Sequence[int], sw_batch_size: int, predictor: Callable, overlap: float = 0.25,
                             mode: str = "constant", padding_mode: str = "constant", cval: float = 0.0, return_labels: bool = False,
                             **unsupported):
    """-> blended predictions (B, K, D, H, W) fp32, or (predictions, uint8 argmax labels (B, 1, D, H, W)) with return_labels.
    Only the arguments the reference drivers use are supported (constant blending / padding); anything else raises."""
    if str(mode).lower() not in ("constant", "blendmode.constant") or str(padding_mode).lower() not in ("constant", "pytorchpadmode.constant"):
        raise NotImplementedError("veloxseg_amd sliding window: constant blending / constant padding only (what utils/inference_*.py use)")
    if unsupported:
        raise NotImplementedError(f"veloxseg_amd sliding window: unsupported arguments {sorted(unsupported)}")
    if inputs.dim() != 5:
        raise ValueError("sliding window expects (B, C, D, H, W)")
    if not inputs.is_cuda:
        raise RuntimeError("veloxseg_amd sliding window runs on MI355X only (no CPU path); move the volume to a cuda device")
    inputs = inputs.contiguous().float()
    B, C = inputs.shape[:2]
    image_size_ = list(inputs.shape[2:])
    roi = tuple(int(r) if r else int(m) for r, m in zip(roi_size, image_size_))
    image_size = tuple(max(m, r) for m, r in zip(image_size_, roi))
    pad = []
    for k in range(4, 1, -1):
        diff = max(roi[k - 2] - inputs.shape[k], 0)
        pad.extend([diff // 2, diff - diff // 2])
    if any(pad):
        inputs = F.pad(inputs, pad=pad, mode="constant", value=cval).contiguous()
    iv = scan_interval(image_size, roi, overlap)
    starts = window_starts(image_size, roi, iv)
    patch = _valid_patch(image_size, roi)
    D, Hh, W = image_size
    rd, rh, rw = patch
    dev = inputs.device
    st = H.stream_ptr
    cz = _axis_counts(sorted({s[0] for s in starts}), rd, D, dev)
    cy = _axis_counts(sorted({s[1] for s in starts}), rh, Hh, dev)
    cx = _axis_counts(sorted({s[2] for s in starts}), rw, W, dev)
    num_win = len(starts)
    total = num_win * B
    acc = None
    K = None
    if (SW_PIPELINE and hasattr(predictor, "forward_slot") and total >= 2 * sw_batch_size and total % sw_batch_size == 0):
        done = _pipelined(inputs, predictor, starts, num_win, total, sw_batch_size, (C, D, Hh, W), (rd, rh, rw))
        if done is not None:
            acc, K = done
            total = 0                                       # (every window batch is in; the loop below has nothing left)
    for g in range(0, total, sw_batch_size):
        idxs = range(g, min(g + sw_batch_size, total))
        win = torch.empty((len(idxs), C, rd, rh, rw), device=dev, dtype=torch.float32)
        for k, i in enumerate(idxs):
            z0, y0, x0 = starts[i % num_win]
            H.call("vx_sw_extract", H.P(inputs[i // num_win]), H.P(win[k]), C, D, Hh, W, rd, rh, rw, z0, y0, x0, st())
        prob = predictor(win)
        if isinstance(prob, (list, tuple)):
            prob = prob[0]
        prob = prob.contiguous().float()
        if acc is None:
            K = prob.shape[1]
            if tuple(prob.shape[2:]) != (rd, rh, rw):
                raise NotImplementedError("predictor output must have the window's spatial size")
            acc = torch.zeros((B, K, D, Hh, W), device=dev, dtype=torch.float32)
        for k, i in enumerate(idxs):
            z0, y0, x0 = starts[i % num_win]
            H.call("vx_sw_accumulate", H.P(prob[k]), H.P(acc[i // num_win]), K, D, Hh, W, rd, rh, rw, z0, y0, x0, 1.0, st())
    labels = torch.empty((B, 1, D, Hh, W), device=dev, dtype=torch.uint8) if return_labels else None
    for b in range(B):
        H.call("vx_sw_finalize", H.P(acc[b]), H.P(acc[b]), H.P(labels[b], torch.uint8) if return_labels else None, H.P(cz), H.P(cy), H.P(cx),
               K, D, Hh, W, st())
    if any(pad):
        lo = [pad[4], pad[2], pad[0]]
        crop = (slice(None), slice(None)) + tuple(slice(l, l + m) for l, m in zip(lo, image_size_))
        acc = acc[crop]
        labels = labels[crop].contiguous() if return_labels else None
    return (acc, labels) if return_labels else acc


def sliding_window_predict(inputs, predictor, roi_size, sw_batch_size, test_config, **kwargs):
    """reference signature (utils/inference_runtime.py:4-19): overlap comes from test_config["sliding_window"]["overlap"]"""
    return sliding_window_inference(inputs, roi_size, sw_batch_size, predictor=predictor, overlap=test_config["sliding_window"]["overlap"], **kwargs)


class Net(torch.nn.Module):
    """forward = first output of the wrapped model (utils/inference_brats.py:41-53 without the Lightning / MONAI data plumbing)"""

    def __init__(self, model: torch.nn.Module):
        super().__init__()
        self._model = model

    def forward(self, x):
        out = self._model(x)
        return out[0] if isinstance(out, (list, tuple)) else out


@torch.inference_mode()
def infer_volume(model: torch.nn.Module, volume: torch.Tensor, roi_size: Sequence[int], sw_batch_size: int, overlap: float, taped: bool = True):
    """eval-mode sliding-window segmentation of one (B, C, D, H, W) volume -> (blended logits, uint8 label map).
    taped (default): the forward of a full window batch is captured once per model and replayed as a launch tape (engine.TapedPredictor); the result
    is the eager forward's (checked at capture).  The accumulator of the blending reads the predictor's static output before the next replay
    overwrites it (same stream).  The predictor hangs on the model object (no process-wide cache, nothing outlives the model) and re-captures
    when the parameters / buffers were re-homed since the capture (TrainEngine's flat buffer, .to(), a parameter swap): a tape bakes in raw
    device pointers."""
    was_training = model.training
    model.eval()
    try:
        net = Net(model) if not isinstance(model, Net) else model
        net.eval()
        pred = net
        if taped and volume.is_cuda and os.environ.get("VELOXSEG_INFER_TAPE", "1") != "0":
            from ..engine import TapedPredictor
            pred = model.__dict__.get("_vx_taped_predictor")
            if pred is None:
                pred = TapedPredictor(net)
                model.__dict__["_vx_taped_predictor"] = pred       # plain attribute (not a sub-module): dies with the model
        return sliding_window_inference(volume, roi_size, sw_batch_size, pred, overlap=overlap, return_labels=True)
    finally:
        model.train(was_training)
