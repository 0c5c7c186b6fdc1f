"""GPU stand-in for the three MONAI dictionary transforms of the reference's training pipeline (utils/train_autopet.py:132-152, utils/train_brats.py):
CropForegroundd(select_fn = x > x.min()), RandCropByPosNegLabeld and RandRotated(range_z) -- same class names, keys and argument meaning, operating on
(C, D, H, W) tensors that are already resident in HBM.  The reductions, the label rank-select and the resampling are HIP kernels (csrc/augment.hip);
random draws are taken on the host from numpy's RandomState in MONAI's order.  There is no CPU fallback."""
import math

import numpy as np
import torch

from .. import _hip as H

_CHUNK = 1 << 16
_INT_MAX = 0x7FFFFFFF


def rotation_range_from_degrees(degrees):
    """reference: utils/runtime.py:115"""
    return math.radians(float(degrees))


def image_label_modes(image_key_count):
    """reference: utils/runtime.py:119"""
    if image_key_count <= 0:
        raise ValueError("image_key_count must be greater than 0")
    return tuple(["bilinear"] * image_key_count + ["nearest"])


def _need_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{what}: the HIP path needs a CUDA (ROCm) tensor; there is no CPU fallback")
    if t.dim() != 4:
        raise ValueError(f"{what}: expected (C, D, H, W), got {tuple(t.shape)}")


def _label_code(lab):
    code = {torch.uint8: 1, torch.int32: 4, torch.int64: 8, torch.float32: -4}.get(lab.dtype)
    if code is None:
        raise TypeError(f"label dtype {lab.dtype} not supported (uint8, int32, int64, float32)")
    return code


def foreground_box(source: torch.Tensor):
    """bounding box (start[3], end[3]) of {source > source.min()} over all channels; zeros when empty"""
    _need_cuda(source, "foreground_box")
    x = source.contiguous().float()
    C, D, Hh, W = x.shape
    st = H.stream_ptr()
    mn = torch.full((1,), float("inf"), device=x.device)
    H.call("vx_min_value", x.data_ptr(), x.numel(), mn.data_ptr(), st)
    box = torch.tensor([_INT_MAX] * 3 + [-1] * 3, dtype=torch.int32, device=x.device)
    H.call("vx_bbox_gt", x.data_ptr(), float(mn.item()), C, D, Hh, W, box.data_ptr(), st)
    b = box.tolist()
    if b[3] < 0:
        return [0, 0, 0], [0, 0, 0]
    return b[:3], [v + 1 for v in b[3:]]


class _LabelIndex:
    """rank-select over the flat label volume: number of foreground / background voxels and the flat index of the k-th one"""

    def __init__(self, label: torch.Tensor):
        _need_cuda(label, "label")
        if label.shape[0] != 1:
            raise ValueError("label must have one channel")
        self.lab = label.contiguous()
        self.code = _label_code(self.lab)
        self.n = self.lab.numel()
        nb = (self.n + _CHUNK - 1) // _CHUNK
        cnt = torch.empty(nb, dtype=torch.int32, device=label.device)
        H.call("vx_label_chunk_count", self.lab.data_ptr(), self.code, self.n, _CHUNK, 1, cnt.data_ptr(), H.stream_ptr())
        fg = cnt.cpu().numpy().astype(np.int64)
        sizes = np.minimum(_CHUNK, self.n - np.arange(nb, dtype=np.int64) * _CHUNK)
        self.cum = {1: np.cumsum(fg), 0: np.cumsum(sizes - fg)}
        self.out = torch.empty(1, dtype=torch.int64, device=label.device)

    def count(self, fg: int) -> int:
        return int(self.cum[fg][-1])

    def kth(self, fg: int, k: int) -> int:
        cum = self.cum[fg]
        ch = int(np.searchsorted(cum, k, side="right"))
        before = int(cum[ch - 1]) if ch else 0
        H.call("vx_label_kth_in_chunk", self.lab.data_ptr(), self.code, self.n, ch * _CHUNK, _CHUNK, fg, k - before, self.out.data_ptr(), H.stream_ptr())
        return int(self.out.item())


def _correct_center(center, spatial_size, shape, allow_smaller):
    spatial_size = list(spatial_size)
    if any(s - p < 0 for s, p in zip(shape, spatial_size)):
        if not allow_smaller:
            raise ValueError("The size of the proposed random crop ROI is larger than the image size")
        spatial_size = [min(s, p) for s, p in zip(shape, spatial_size)]
    out = []
    for c, p, s in zip(center, spatial_size, shape):
        lo = p // 2
        hi = int(s + 1 - p / 2.0)
        if hi == lo:
            hi += 1
        out.append(int(min(max(c, lo), hi - 1)))
    return out


class _Randomizable:
    def set_random_state(self, seed=None, state=None):
        self.R = state if state is not None else np.random.RandomState(seed)
        return self


class CropForegroundd:
    def __init__(self, keys, source_key, select_fn=None):
        self.keys, self.source_key = list(keys), source_key
        if select_fn is not None:      # only the reference's selector (x > x.min()) runs on the GPU: check that this is what was passed
            for probe in (torch.tensor([[-3.0, -3.0, 5.0, -1.0]]), torch.tensor([[10.0, 11.0, 10.5, 10.0]])):
                if torch.equal(torch.as_tensor(select_fn(probe)), probe > probe.min()):
                    continue
                raise NotImplementedError("CropForegroundd: only select_fn = (x > x.min()) is implemented on the HIP path")

    def __call__(self, data):
        start, end = foreground_box(data[self.source_key])
        out = dict(data)
        for k in self.keys:
            out[k] = data[k][:, start[0]:end[0], start[1]:end[1], start[2]:end[2]]
        out["foreground_start_coord"], out["foreground_end_coord"] = start, end
        return out


class RandCropByPosNegLabeld(_Randomizable):
    def __init__(self, keys, label_key, spatial_size, pos=1.0, neg=1.0, num_samples=1, allow_smaller=False):
        if pos < 0 or neg < 0 or pos + neg == 0:
            raise ValueError("pos and neg must be nonnegative and not both zero")
        self.keys, self.label_key, self.spatial_size = list(keys), label_key, list(spatial_size)
        self.pos_ratio, self.num_samples, self.allow_smaller = pos / (pos + neg), int(num_samples), allow_smaller
        self.set_random_state()

    def centers(self, label):
        index = _LabelIndex(label)
        shape = tuple(label.shape[1:])
        nfg, nbg = index.count(1), index.count(0)
        ratio = self.pos_ratio
        if nfg == 0 and nbg == 0:
            raise ValueError("No sampling location available.")
        if nfg == 0 or nbg == 0:
            ratio = 0 if nfg == 0 else 1
        out = []
        for _ in range(self.num_samples):
            fg = 1 if self.R.rand() < ratio else 0
            flat = index.kth(fg, int(self.R.randint(nfg if fg else nbg)))
            out.append(_correct_center(list(np.unravel_index(flat, shape)), self.spatial_size, shape, self.allow_smaller))
        return out

    def __call__(self, data):
        shape = tuple(data[self.label_key].shape[1:])
        res = []
        for c in self.centers(data[self.label_key]):
            sl = tuple(slice(max(ci - p // 2, 0), min(max(ci - p // 2, 0) + p, s)) for ci, p, s in zip(c, self.spatial_size, shape))
            d = dict(data)
            for k in self.keys:
                d[k] = data[k][(slice(None),) + sl].contiguous()
            d["crop_center"] = c
            res.append(d)
        return res


def rotate_z(x: torch.Tensor, angle: float, mode: str = "bilinear") -> torch.Tensor:
    _need_cuda(x, "rotate_z")
    if mode not in ("bilinear", "nearest"):
        raise ValueError(f"mode {mode!r} not supported")
    xin = x.contiguous().float()
    out = torch.empty_like(xin)
    C, D, Hh, W = xin.shape
    H.call("vx_rotate_z", xin.data_ptr(), out.data_ptr(), C, D, Hh, W, math.cos(angle), math.sin(angle), 0 if mode == "bilinear" else 1, H.stream_ptr())
    return out.to(x.dtype) if x.dtype != torch.float32 else out


class RandRotated(_Randomizable):
    def __init__(self, keys, range_x=0.0, range_y=0.0, range_z=0.0, prob=0.1, mode="bilinear", padding_mode="border"):
        if range_x or range_y:
            raise NotImplementedError("RandRotated: only rotation about z (range_z) is implemented on the HIP path (the reference uses no other)")
        if padding_mode != "border":
            raise NotImplementedError("RandRotated: only padding_mode='border' (MONAI's default) is implemented")
        self.keys, self.range_z, self.prob = list(keys), float(range_z), float(prob)
        self.mode = [mode] * len(self.keys) if isinstance(mode, str) else list(mode)
        if len(self.mode) != len(self.keys):
            raise ValueError("mode must have one entry per key")
        self.set_random_state()

    def draw(self):
        do = self.R.rand() < self.prob
        self.R.uniform(low=-0.0, high=0.0)
        self.R.uniform(low=-0.0, high=0.0)
        return do, float(self.R.uniform(low=-self.range_z, high=self.range_z))

    def __call__(self, data):
        if isinstance(data, list):
            return [self(d) for d in data]
        do, angle = self.draw()
        out = dict(data)
        if do:
            for k, m in zip(self.keys, self.mode):
                out[k] = rotate_z(data[k], angle, m)
        return out


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def set_random_state(self, seed=None):
        rs = np.random.RandomState(seed)
        for t in self.transforms:
            if isinstance(t, _Randomizable):
                t.set_random_state(seed=int(rs.randint(2 ** 31 - 1)))
        return self

    def __call__(self, data):
        for t in self.transforms:
            data = [t(d) for d in data] if isinstance(data, list) and not isinstance(t, RandRotated) else t(data)
        return data
