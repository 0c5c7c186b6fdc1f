"""Per-sample confusion matrix of two label volumes on the GPU (csrc/infer.hip vx_confusion): the only device work the reference's
metrics need; 8 * NC^2 bytes come back to the host instead of the whole volume (utils/metric/metrics.py:61-62 copies both volumes)."""
import torch

from ... import _hip as H

_BYTES = {torch.uint8: 1, torch.int32: 4, torch.int64: 8}


def to_label_map(t: torch.Tensor) -> torch.Tensor:
    """(B, 1, ...) integer labels stay; (B, K, ...) float scores are arg-maxed over K on the GPU (uint8)"""
    if t.is_floating_point():
        if t.shape[1] == 1:
            return t.to(torch.int64)
        B, K = t.shape[:2]
        V = t[0, 0].numel()
        out = torch.empty((B, 1) + tuple(t.shape[2:]), device=t.device, dtype=torch.uint8)
        H.call("vx_argmax_channels", H.P(t.contiguous().float()), H.P(out, torch.uint8), B, K, V, H.stream_ptr())
        return out
    return t


def confusion(pred: torch.Tensor, gt: torch.Tensor, num_classes: int) -> torch.Tensor:
    """-> int64 (B, NC, NC) on the HOST: conf[b, g, p] = #voxels of sample b with ground truth g predicted as p"""
    if not (pred.is_cuda and gt.is_cuda):
        raise RuntimeError("veloxseg_amd metrics run on MI355X only (no CPU path)")
    if pred.shape != gt.shape:
        raise ValueError(f"prediction {tuple(pred.shape)} and label {tuple(gt.shape)} shapes differ")
    pred = pred.contiguous()
    gt = gt.contiguous()
    for t in (pred, gt):
        if t.dtype not in _BYTES:
            raise TypeError(f"label dtype {t.dtype} not supported (uint8 / int32 / int64)")
    if _BYTES[pred.dtype] == 4 and _BYTES[gt.dtype] != 4 or _BYTES[pred.dtype] == 8 and _BYTES[gt.dtype] == 4:
        gt = gt.to(pred.dtype)
    B = pred.shape[0]
    V = pred[0].numel()
    conf = torch.zeros((B, num_classes, num_classes), device=pred.device, dtype=torch.int64)
    H.call("vx_confusion", H.P(pred, None), _BYTES[pred.dtype], H.P(gt, None), _BYTES[gt.dtype], H.P(conf, torch.int64), B, V, num_classes, H.stream_ptr())
    return conf.cpu()
