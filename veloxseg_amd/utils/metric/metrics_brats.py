"""BraTS region metrics -- mirrors utils/metric/metrics_brats.py (ET = label 3, TC = {1, 3}, WT = non-zero; Dice with eps 1e-6).

The reference's `Dice(output, target)` sums over dims (1, 2, 3) of (B, 1, D, H, W) masks, i.e. over (channel, D, H) -- it keeps the
LAST spatial axis -- and averages `2 (inter + eps) / (sum_o + sum_t + 2 eps)` over all B*W planes (metrics_brats.py:22-27).  That
per-plane statistic is reproduced exactly: one confusion matrix per (sample, last-axis index) on the GPU, ratio / mean in float32 on the host.
"""
import torch

from ._counts import confusion, to_label_map


def _planes(t: torch.Tensor) -> torch.Tensor:
    """(B, 1, D, H, W) -> (B*W, D*H): the voxels that share a last-axis index are one row"""
    if t.dim() != 5 or t.shape[1] != 1:
        raise ValueError("BraTS Dice expects (B, 1, D, H, W) label maps")
    B, _, D, Hh, W = t.shape
    return t.permute(0, 4, 1, 2, 3).reshape(B * W, D * Hh)


def _slice_conf(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """(B, 1, D, H, W) label maps -> (B*W, 4, 4) confusion matrices"""
    return confusion(_planes(output), _planes(target), 4)


def _dice_from(inter, so, st, eps=1e-6):
    inter = inter.float() + eps
    union = so.float() + st.float() + eps * 2
    return torch.mean(2 * inter / union)


def cal_dice(output: torch.Tensor, target: torch.Tensor):
    """-> (avg, ET, TC, WT) floats; argument order as in the reference (it is symmetric)"""
    conf = _slice_conf(to_label_map(output), to_label_map(target))       # conf[s, target class, output class]
    po = conf.sum(1)             # per output class
    pt = conf.sum(2)             # per target class
    et = _dice_from(conf[:, 3, 3], po[:, 3], pt[:, 3])
    tc_i = conf[:, 1, 1] + conf[:, 1, 3] + conf[:, 3, 1] + conf[:, 3, 3]
    tc = _dice_from(tc_i, po[:, 1] + po[:, 3], pt[:, 1] + pt[:, 3])
    wt_i = conf[:, 1:, 1:].sum((1, 2))
    wt = _dice_from(wt_i, po[:, 1:].sum(1), pt[:, 1:].sum(1))
    return float((et + tc + wt) / 3), float(et), float(tc), float(wt)


def Dice(output: torch.Tensor, target: torch.Tensor, eps=1e-6):
    """binary-mask Dice of metrics_brats.py:22-27 (masks as 0/1 floats or ints, (B, 1, D, H, W))"""
    o = (output != 0).to(torch.uint8)
    t = (target != 0).to(torch.uint8)
    conf = confusion(_planes(o), _planes(t), 2)
    return _dice_from(conf[:, 1, 1], conf[:, :, 1].sum(1), conf[:, 1, :].sum(1), eps)


def show_deep_metrics(outputs, labels, deep=True):
    """metrics_brats.py:6-20"""
    if not isinstance(outputs, (list, tuple)):
        outputs = [outputs]
    res, string = None, ""
    for k, o in enumerate(outputs if deep else outputs[:1]):
        out = to_label_map(o)
        avg, et, tc, wt = cal_dice(labels, out)
        string += f"[Avg:{avg:.4f}, ET:{et:.4f}, TC:{tc:.4f}, WT:{wt:.4f} pix:{int((out != 0).sum()):6}/{int((labels != 0).sum()):6}]\n"
        if k == 0:
            res = [avg, et, tc, wt]
    return res, string + "\n"
