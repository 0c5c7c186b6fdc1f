"""Host logic of the input-pipeline stand-in + the oracle's own internal properties (no GPU)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import augment_oracle as AO            # noqa: E402
from veloxseg_amd.utils import augment as A        # noqa: E402


def test_reference_helpers():
    assert A.rotation_range_from_degrees(15) == pytest.approx(math.pi / 12)
    assert A.image_label_modes(2) == ("bilinear", "bilinear", "nearest")
    with pytest.raises(ValueError):
        A.image_label_modes(0)


def test_center_correction_matches_oracle():
    rs = np.random.RandomState(0)
    for _ in range(200):
        shape = tuple(int(v) for v in rs.randint(8, 40, 3))
        size = [int(rs.randint(1, s + 1)) for s in shape]
        c = [int(rs.randint(0, s)) for s in shape]
        got = A._correct_center(c, size, shape, False)
        assert got == AO.correct_center(c, size, shape, False)
        sl = AO.crop_slices(got, size, shape)
        assert all(s.stop - s.start == p for s, p in zip(sl, size)), (shape, size, c, got)      # a corrected centre always yields a full-size crop
    with pytest.raises(ValueError):
        A._correct_center([1, 1, 1], [9, 4, 4], (8, 8, 8), False)
    assert A._correct_center([1, 1, 1], [9, 4, 4], (8, 8, 8), True) == AO.correct_center([1, 1, 1], [9, 4, 4], (8, 8, 8), True)


def test_oracle_rotation_properties():
    x = torch.randn(2, 9, 11, 5)
    assert torch.allclose(AO.rotate_z(x, 0.0, "bilinear"), x, atol=1e-6)
    assert torch.equal(AO.rotate_z(x, 0.0, "nearest"), x)
    # a quarter turn of a square plane is an exact permutation: out[d, h] = in[c - (h - c), d] ...
    y = torch.randn(1, 7, 7, 3)
    r = AO.rotate_z(y, math.pi / 2, "nearest")
    assert torch.allclose(r, torch.rot90(y, 1, dims=(1, 2)), atol=1e-6) or torch.allclose(r, torch.rot90(y, -1, dims=(1, 2)), atol=1e-6)
    assert torch.allclose(AO.rotate_z(r, -math.pi / 2, "nearest"), y, atol=1e-6)


def test_oracle_bounding_box_and_indices():
    a = np.zeros((2, 6, 7, 8), np.float32)
    a[1, 2:4, 1:6, 3] = 1.0
    assert AO.bounding_box(a) == ([2, 1, 3], [4, 6, 4])
    assert AO.bounding_box(np.ones((1, 3, 3, 3), np.float32)) == ([0, 0, 0], [0, 0, 0])
    lab = np.zeros((1, 4, 4, 4), np.uint8)
    lab[0, 1, 2, 3] = 2
    fg, bg = AO.fg_bg_indices(lab)
    assert fg.tolist() == [1 * 16 + 2 * 4 + 3] and len(bg) == 63


def test_product_refuses_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.foreground_box(torch.zeros(1, 4, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.rotate_z(torch.zeros(1, 4, 4, 4), 0.1)
    with pytest.raises(NotImplementedError):
        A.CropForegroundd(["img"], "img", select_fn=lambda x: x > 0.5)
    A.CropForegroundd(["img"], "img", select_fn=lambda x: x > x.min())
    with pytest.raises(NotImplementedError):
        A.RandRotated(["img"], range_x=0.1)
