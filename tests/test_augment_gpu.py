"""Input-pipeline stand-in: HIP kernels vs the CPU oracle on the same inputs and the same random draws."""
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

pytestmark = pytest.mark.gpu


def _volume(seed, shape=(1, 70, 61, 53)):
    g = torch.Generator().manual_seed(seed)
    x = torch.full(shape, -1.5)
    C, D, Hh, W = shape
    x[:, 9:D - 12, 7:Hh - 11, 11:W - 13] = torch.randn(C, D - 21, Hh - 18, W - 24, generator=g).abs() - 1.4
    return x


@pytest.mark.parametrize("seed", [0, 1])
def test_foreground_box_matches_oracle(seed):
    x = _volume(seed)
    s, e = A.foreground_box(x.cuda())
    assert (s, e) == AO.bounding_box(x.numpy())
    assert A.foreground_box(torch.ones(2, 5, 6, 7).cuda()) == ([0, 0, 0], [0, 0, 0])


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int32, torch.int64, torch.float32])
def test_crop_centers_bit_exact(dtype):
    g = torch.Generator().manual_seed(3)
    lab = (torch.rand(1, 90, 75, 67, generator=g) > 0.97).to(dtype)            # 454k voxels -> 7 chunks, ragged last one
    t = A.RandCropByPosNegLabeld(["seg"], "seg", [32, 32, 32], pos=1, neg=1, num_samples=16).set_random_state(seed=11)
    got = t.centers(lab.cuda())
    want = AO.crop_centers(lab.numpy(), [32, 32, 32], 16, 1, 1, np.random.RandomState(11))
    assert got == want


def test_crop_all_background_and_all_foreground():
    t = A.RandCropByPosNegLabeld(["seg"], "seg", [8, 8, 8], num_samples=4).set_random_state(seed=2)
    lab = torch.zeros(1, 20, 20, 20, dtype=torch.uint8)
    assert t.centers(lab.cuda()) == AO.crop_centers(lab.numpy(), [8, 8, 8], 4, 1, 1, np.random.RandomState(2))
    t.set_random_state(seed=4)
    lab = torch.ones(1, 20, 20, 20, dtype=torch.uint8)
    assert t.centers(lab.cuda()) == AO.crop_centers(lab.numpy(), [8, 8, 8], 4, 1, 1, np.random.RandomState(4))
    with pytest.raises(ValueError, match="larger than the image"):
        A.RandCropByPosNegLabeld(["seg"], "seg", [32, 8, 8]).centers(lab.cuda())


@pytest.mark.parametrize("angle", [0.0, 0.1234, -math.radians(15), 0.7])
def test_rotation_matches_oracle(angle):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 40, 33, 21, generator=g)
    got = A.rotate_z(x.cuda(), angle, "bilinear").cpu()
    assert torch.allclose(got, AO.rotate_z(x, angle, "bilinear"), atol=2e-4), float((got - AO.rotate_z(x, angle, "bilinear")).abs().max())
    lab = (torch.rand(1, 40, 33, 21, generator=g) > 0.8).float()
    gn = A.rotate_z(lab.cuda(), angle, "nearest").cpu()
    assert set(gn.unique().tolist()) <= {0.0, 1.0}
    mism = float((gn != AO.rotate_z(lab, angle, "nearest")).float().mean())       # exact .5 ties may round differently in fp32 vs fp64 coordinates
    assert mism < 2e-3, mism
    if angle == 0.0:
        assert torch.equal(got, x) and torch.equal(gn, lab)


def test_pipeline_end_to_end_against_oracle():
    """the reference's transform chain on one synthetic case: same patches as the oracle taking the same draws"""
    x = _volume(7, (1, 96, 80, 72))
    ct = _volume(8, (1, 96, 80, 72))
    seg = torch.zeros(1, 96, 80, 72)
    seg[:, 30:40, 30:44, 30:43] = 1.0
    keys = ["img", "img_ct", "seg"]
    crop = A.RandCropByPosNegLabeld(keys, "seg", [32, 32, 32], pos=1, neg=1, num_samples=2).set_random_state(seed=21)
    rot = A.RandRotated(keys, range_z=A.rotation_range_from_degrees(15), mode=A.image_label_modes(2), prob=0.5).set_random_state(seed=22)
    pipe = A.Compose([A.CropForegroundd(keys, "img", select_fn=lambda v: v > v.min()), crop, rot])
    got = pipe({"img": x.cuda(), "img_ct": ct.cuda(), "seg": seg.cuda()})
    # oracle chain
    s, e = AO.bounding_box(x.numpy())
    cr = {k: v[:, s[0]:e[0], s[1]:e[1], s[2]:e[2]] for k, v in (("img", x), ("img_ct", ct), ("seg", seg))}
    centers = AO.crop_centers(cr["seg"].numpy(), [32, 32, 32], 2, 1, 1, np.random.RandomState(21))
    rs = np.random.RandomState(22)
    assert len(got) == 2
    for d, c in zip(got, centers):
        sl = (slice(None),) + AO.crop_slices(c, [32, 32, 32], cr["seg"].shape[1:])
        do, ang = AO.rand_rotate_draw(rs, math.radians(15), 0.5)
        for k, m in zip(keys, ("bilinear", "bilinear", "nearest")):
            want = cr[k][sl]
            if do:
                want = AO.rotate_z(want.contiguous(), ang, m)
            assert d[k].shape == (1, 32, 32, 32)
            if m == "bilinear":
                assert torch.allclose(d[k].cpu(), want, atol=2e-4)
            else:
                assert float((d[k].cpu() != want).float().mean()) < 2e-3
