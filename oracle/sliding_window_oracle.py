"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the sliding-window inference used by the reference's test drivers.

The reference delegates to a third-party dependency that is NOT vendored under /root/reference:
    utils/inference_runtime.py:4-19  ->  monai.inferers.sliding_window_inference   (monai==1.5.0, requirements.txt:3)
    call sites: utils/inference_brats.py:209-215, utils/inference_petct.py (same pattern); followed by argmax(dim=1) (:216).
MONAI is absent from this image, so this file restates its published algorithm (monai/inferers/utils.py `sliding_window_inference`,
`_get_scan_interval`; monai/data/utils.py `dense_patch_slices`, `get_valid_patch_size`, `compute_importance_map` constant mode) for the
argument subset the reference uses (overlap from test_config, everything else default: constant blending, constant padding 0).
PARITY UNPINNED against MONAI itself (no vectors from the real library can be produced here); pinned instead by
  * the window counts SURVEY.md 8d derives for BASELINE config 5 (240x240x155: roi 96 / overlap .5 -> 4x4x3 = 48, roi 128 -> 18, overlap .25 -> 18),
  * exactness properties (a position-wise predictor must be reproduced exactly; windows cover every voxel).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from typing import Callable, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def get_valid_patch_size(image_size: Sequence[int], patch_size: Sequence[int]) -> Tuple[int, ...]:
    """monai/data/utils.py: patch no larger than the image; 0 / None entries fall back to the image size"""
    return tuple(min(m, p) if p else m for m, p in zip(image_size, patch_size))


def get_scan_interval(image_size, roi_size, overlap) -> Tuple[int, ...]:
    """monai/inferers/utils.py `_get_scan_interval`: int(roi * (1 - overlap)), at least 1; the full roi when it spans the axis"""
    out = []
    for i in range(len(image_size)):
        if roi_size[i] == image_size[i]:
            out.append(int(roi_size[i]))
        else:
            interval = int(roi_size[i] * (1 - overlap))
            out.append(interval if interval > 0 else 1)
    return tuple(out)


def dense_patch_starts(image_size, patch_size, scan_interval) -> List[Tuple[int, ...]]:
    """monai/data/utils.py `dense_patch_slices`: window origins, first axis slowest; the last window of an axis is clamped to the edge"""
    nd = len(image_size)
    patch_size = get_valid_patch_size(image_size, patch_size)
    scan_num = []
    for i in range(nd):
        if scan_interval[i] == 0:
            scan_num.append(1)
        else:
            num = int(math.ceil(float(image_size[i]) / scan_interval[i]))
            scan_dim = next((d for d in range(num) if d * scan_interval[i] + patch_size[i] >= image_size[i]), None)
            scan_num.append(scan_dim + 1 if scan_dim is not None else 1)
    starts = []
    for dim in range(nd):
        dim_starts = []
        for idx in range(scan_num[dim]):
            s = idx * scan_interval[dim]
            s -= max(s + patch_size[dim] - image_size[dim], 0)
            dim_starts.append(s)
        starts.append(dim_starts)
    out = np.asarray([x.flatten() for x in np.meshgrid(*starts, indexing="ij")]).T
    return [tuple(int(v) for v in row) for row in out]


def sliding_window_inference(inputs: torch.Tensor, roi_size: Sequence[int], sw_batch_size: int, predictor: Callable,
                             overlap: float = 0.25, cval: float = 0.0) -> torch.Tensor:
    """constant blending: out = (sum over windows of predictor(window)) / (number of windows covering the voxel)"""
    nd = inputs.dim() - 2
    batch_size = inputs.shape[0]
    image_size_ = list(inputs.shape[2:])
    roi_size = tuple(int(r) if r else int(m) for r, m in zip(roi_size, image_size_))          # fall_back_tuple
    image_size = tuple(max(image_size_[i], roi_size[i]) for i in range(nd))
    pad_size = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi_size[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad_size.extend([half, diff - half])
    if any(pad_size):
        inputs = F.pad(inputs, pad=pad_size, mode="constant", value=cval)
    interval = get_scan_interval(image_size, roi_size, overlap)
    starts = dense_patch_starts(image_size, roi_size, interval)
    patch = get_valid_patch_size(image_size, roi_size)
    num_win = len(starts)
    total = num_win * batch_size
    out = None
    count = None
    for g in range(0, total, sw_batch_size):
        idxs = list(range(g, min(g + sw_batch_size, total)))
        sl = [(slice(i // num_win, i // num_win + 1), slice(None)) + tuple(slice(s, s + p) for s, p in zip(starts[i % num_win], patch)) for i in idxs]
        win = torch.cat([inputs[s] for s in sl]) if len(sl) > 1 else inputs[sl[0]]
        prob = predictor(win)
        if isinstance(prob, (list, tuple)):
            prob = prob[0]
        if out is None:
            out = torch.zeros((batch_size, prob.shape[1]) + tuple(image_size), dtype=inputs.dtype, device=prob.device)
            count = torch.zeros((1, 1) + tuple(image_size), dtype=inputs.dtype, device=prob.device)
            for st in starts:
                count[(slice(None), slice(None)) + tuple(slice(s, s + p) for s, p in zip(st, patch))] += 1.0
        for k, s in enumerate(sl):
            out[s] += 1.0 * prob[k:k + 1]
    out = out / count
    if any(pad_size):
        crop = [slice(None), slice(None)]
        for d in range(nd):
            lo = pad_size[2 * (nd - 1 - d)]
            crop.append(slice(lo, lo + image_size_[d]))
        out = out[tuple(crop)]
    return out
