"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy + torch CPU) of the three MONAI dictionary transforms the reference's training pipeline applies
between loading and the model (utils/train_autopet.py:132-152, utils/train_brats.py same block): CropForegroundd(select_fn = x > x.min()),
RandCropByPosNegLabeld(pos=1, neg=1, num_samples=2) and RandRotated(range_z, bilinear image / nearest label, border padding, prob 0.5).

MONAI (requirements.txt: monai==1.5.0) is a third-party dependency that is absent from this image and from /root/reference, so this restates its
published algorithm: PARITY UNPINNED against MONAI itself.  What is pinned: the reference's own helpers around it (rotation_range_from_degrees,
image_label_modes: utils/runtime.py:115-122) and internal properties (tests/test_augment_*.py).

Restated from MONAI 1.5.0:
  transforms/utils.py generate_spatial_bounding_box      -> bounding_box
  transforms/utils.py map_binary_to_indices              -> fg_bg_indices
  transforms/utils.py generate_pos_neg_label_crop_centers / correct_crop_centers -> crop_centers / correct_center
  transforms/croppad/array.py SpatialCrop(roi_center, roi_size) -> crop_slices
  transforms/spatial/array.py Rotate (create_rotate about z, shift to the (n-1)/2 centre, AffineTransform normalized=False, padding "border") -> rotate_z
  transforms/spatial/array.py RandRotate.randomize (draw order: do-flag, x, y, z) -> rand_rotate_draw
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def bounding_box(source: np.ndarray):
    """source (C, D, H, W); select x > x.min() over the whole array, any over channels; returns (start[3], end[3]) (end exclusive; zeros when nothing selected)"""
    sel = (source > source.min()).any(0)
    nd = sel.ndim
    if not sel.any():
        return [0] * nd, [0] * nd
    start, end = [], []
    for ax in range(nd):
        other = tuple(a for a in range(nd) if a != ax)
        hit = np.nonzero(sel.any(axis=other))[0]
        start.append(int(hit[0]))
        end.append(int(hit[-1]) + 1)
    return start, end


def fg_bg_indices(label: np.ndarray):
    flat = (label > 0).any(0).ravel()
    return np.nonzero(flat)[0], np.nonzero(~flat)[0]


def correct_center(center, spatial_size, shape, allow_smaller=False):
    spatial_size = list(spatial_size)
    if any(s - p < 0 for s, p in zip(shape, spatial_size)):
        if not allow_smaller:
            raise ValueError("The size of the proposed random crop ROI is larger than the image size")
        spatial_size = [min(s, p) for s, p in zip(shape, spatial_size)]
    valid_start = np.floor_divide(spatial_size, 2)
    valid_end = np.subtract(np.asarray(shape) + 1, np.asarray(spatial_size) / 2.0).astype(np.uint16)
    for i, vs in enumerate(valid_start):
        if vs == valid_end[i]:
            valid_end[i] += 1
    return [int(min(max(c, vs), ve - 1)) for c, vs, ve in zip(center, valid_start, valid_end)]


def crop_centers(label: np.ndarray, spatial_size, num_samples, pos, neg, rs: np.random.RandomState, allow_smaller=False):
    fg, bg = fg_bg_indices(label)
    shape = label.shape[1:]
    pos_ratio = pos / (pos + neg)
    if len(fg) == 0 and len(bg) == 0:
        raise ValueError("No sampling location available.")
    if len(fg) == 0 or len(bg) == 0:
        pos_ratio = 0 if len(fg) == 0 else 1
    out = []
    for _ in range(num_samples):
        use = fg if rs.rand() < pos_ratio else bg
        idx = use[rs.randint(len(use))]
        out.append(correct_center(list(np.unravel_index(idx, shape)), spatial_size, shape, allow_smaller))
    return out


def crop_slices(center, spatial_size, shape):
    sl = []
    for c, p, s in zip(center, spatial_size, shape):
        a = max(c - p // 2, 0)
        sl.append(slice(a, min(a + p, s)))
    return tuple(sl)


def rand_rotate_draw(rs: np.random.RandomState, range_z, prob):
    do = rs.rand() < prob
    rs.uniform(low=-0.0, high=0.0)                 # x
    rs.uniform(low=-0.0, high=0.0)                 # y
    z = rs.uniform(low=-range_z, high=range_z)
    return do, float(z)


def rotate_z(x: torch.Tensor, angle: float, mode: str) -> torch.Tensor:
    """x (C, D, H, W) on the CPU; output voxel p takes the input at R (p - c) + c with c = (n-1)/2 in the (D, H) plane; border padding"""
    C, D, H, W = x.shape
    cs, sn = math.cos(angle), math.sin(angle)
    d = torch.arange(D, dtype=torch.float64) - (D - 1) / 2
    h = torch.arange(H, dtype=torch.float64) - (H - 1) / 2
    dd, hh = torch.meshgrid(d, h, indexing="ij")
    sd = cs * dd - sn * hh + (D - 1) / 2
    sh = sn * dd + cs * hh + (H - 1) / 2
    # grid_sample wants (x, y, z) = (W, H, D) order, normalised with align_corners=True: u = 2 p / (n - 1) - 1
    gw = torch.arange(W, dtype=torch.float64) * (2.0 / max(W - 1, 1)) - 1
    grid = torch.empty(1, D, H, W, 3, dtype=torch.float64)
    grid[..., 0] = gw.view(1, 1, 1, W)
    grid[..., 1] = (sh * (2.0 / max(H - 1, 1)) - 1).view(1, D, H, 1)
    grid[..., 2] = (sd * (2.0 / max(D - 1, 1)) - 1).view(1, D, H, 1)
    out = F.grid_sample(x.double()[None], grid, mode=mode, padding_mode="border", align_corners=True)
    return out[0].to(x.dtype)
