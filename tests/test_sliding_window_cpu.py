"""CPU: sliding-window geometry of the product's host code against the oracle restatement and the window counts SURVEY.md 8d derives for
BASELINE config 5; oracle exactness properties.  (No GPU: only the pure-python window planner of veloxseg_amd is imported.)"""
import random

import pytest
import torch

from oracle import sliding_window_oracle as SO
from veloxseg_amd.utils import inference_runtime as IR


@pytest.mark.parametrize("roi,overlap,count", [(96, 0.5, 48), (128, 0.5, 18), (96, 0.25, 18)])
def test_brats_volume_window_counts(roi, overlap, count):
    image = (240, 240, 155)
    for mod in (SO, None):
        if mod is SO:
            iv = SO.get_scan_interval(image, (roi,) * 3, overlap)
            starts = SO.dense_patch_starts(image, (roi,) * 3, iv)
        else:
            iv = IR.scan_interval(image, (roi,) * 3, overlap)
            starts = IR.window_starts(image, (roi,) * 3, iv)
        assert len(starts) == count
        assert iv == (int(roi * (1 - overlap)),) * 3
        assert max(s[2] for s in starts) == 155 - roi and max(s[0] for s in starts) == 240 - roi      # last window clamped to the edge


def test_planner_matches_oracle_on_random_geometries():
    rng = random.Random(7)
    for _ in range(300):
        image = tuple(rng.randint(1, 40) for _ in range(3))
        roi = tuple(min(rng.randint(1, 24), m) for m in image)
        overlap = rng.choice([0.0, 0.25, 0.5, 0.75, 0.9])
        iv_o = SO.get_scan_interval(image, roi, overlap)
        assert IR.scan_interval(image, roi, overlap) == iv_o
        st_o = SO.dense_patch_starts(image, roi, iv_o)
        st_p = IR.window_starts(image, roi, iv_o)
        assert st_p == st_o
        cover = torch.zeros(image)
        for s in st_o:
            assert all(0 <= a and a + r <= m for a, r, m in zip(s, roi, image))
            cover[s[0]:s[0] + roi[0], s[1]:s[1] + roi[1], s[2]:s[2] + roi[2]] += 1
        assert float(cover.min()) >= 1, "every voxel is covered by at least one window"


@pytest.mark.parametrize("shape,roi,swb,overlap", [((2, 3, 20, 17, 9), (8, 8, 8), 3, 0.5), ((1, 2, 5, 6, 7), (8, 4, 16), 2, 0.25), ((1, 1, 12, 12, 12), (12, 12, 12), 4, 0.5)])
def test_oracle_reproduces_a_pointwise_predictor(shape, roi, swb, overlap):
    """constant blending of identical per-voxel values is the value itself (up to the rounding of sum / count), including the padded case"""
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1))
    out = SO.sliding_window_inference(x, roi, swb, lambda w: torch.cat([w[:, :1] * 2.0, w[:, :1] - 1.0], 1), overlap)
    want = torch.cat([x[:, :1] * 2.0, x[:, :1] - 1.0], 1)
    assert out.shape == want.shape
    # windows that hang over the padded border see cval = 0 there, but padded voxels are cropped away again
    assert float((out - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))
