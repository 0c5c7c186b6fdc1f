"""Seeded, construction-order-independent recipes for golden inputs (this repo's own code).

The golden generator (make_golden.py, run once against the imported reference) and the parity tests
both call these, so multi-MB state dicts and inputs need not be committed: the fixtures hold only
expected OUTPUTS plus sha256 checksums of the regenerated inputs (a checksum mismatch = RNG drift,
reported loudly instead of as a numeric mismatch).
"""
import hashlib
import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_state_dict(sd: dict, seed: int = 7) -> dict:
    """Return a new state dict with every floating tensor replaced by a value that depends only on
    (name, shape, seed).  Integer buffers (relative_position_index) are kept as constructed."""
    out = {}
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        g = _gen(k, seed)
        r = torch.randn(v.shape, generator=g, dtype=torch.float32)
        if k.endswith("relative_position_bias_table"):
            t = 0.5 * r
        elif v.ndim == 1 and k.endswith("weight"):
            t = 1.0 + 0.1 * r           # LayerNorm scales
        elif v.ndim == 1:
            t = 0.1 * r                 # biases
        else:
            fan_in = v.numel() // v.shape[0]
            t = r * (2.0 / fan_in) ** 0.5
        out[k] = t.to(v.dtype)
    return out


def make_inputs(cfg: dict, batch: int, seed: int = 2024):
    g = torch.Generator().manual_seed(seed)
    size = list(cfg["input_size"])
    x = torch.randn(batch, sum(cfg["in_ch"]), *size, generator=g)
    ncls = cfg["n_classes"]
    if ncls == 2:
        labels = (torch.rand(batch, 1, *size, generator=g) > 0.9).long()
    else:  # blocky multi-class labels
        labels = torch.randint(0, ncls, (batch, 1, *[s // 8 for s in size]), generator=g)
        labels = labels.repeat_interleave(8, 2).repeat_interleave(8, 3).repeat_interleave(8, 4)
    return x, labels


def tensor_sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def sd_sha(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def compact(t: torch.Tensor, limit: int = 70000) -> dict:
    """Small tensors in full; big ones as shape + sums + a strided spatial subsample (stride 3; coarser -- recorded as "stride" -- for the
    128^3 cases so that a fixture stays well under 2 MB)."""
    t = t.detach()
    if t.numel() <= limit:
        return {"full": t.clone()}
    st = 3
    while t[..., ::st, ::st, ::st].numel() > 45000:
        st += 1
    out = {"shape": list(t.shape), "sum": float(t.double().sum()), "abs_sum": float(t.double().abs().sum()),
           "strided": t[..., ::st, ::st, ::st].clone()}
    if st != 3:
        out["stride"] = st
    return out


def pack_mask(am: torch.Tensor, ncls: int) -> dict:
    """arg-max mask (uint8, values < ncls <= 4) as bit planes: 1/8 (2 classes) or 1/4 (<= 4 classes) of a byte per voxel"""
    import numpy as np
    a = am.detach().cpu().numpy().astype(np.uint8)
    planes = [np.packbits((a >> k) & 1) for k in range(1 if ncls <= 2 else 2)]
    return {"shape": list(a.shape), "planes": [torch.from_numpy(p.copy()) for p in planes]}


def unpack_mask(d) -> torch.Tensor:
    import numpy as np
    if torch.is_tensor(d):
        return d
    n = 1
    for s in d["shape"]:
        n *= s
    out = np.zeros(n, dtype=np.uint8)
    for k, p in enumerate(d["planes"]):
        out |= (np.unpackbits(p.numpy())[:n] << k).astype(np.uint8)
    return torch.from_numpy(out.reshape(d["shape"]))


def check_compact(t: torch.Tensor, ref: dict, atol: float, rtol: float, what: str = ""):
    t = t.detach().cpu().float()
    if "full" in ref:
        torch.testing.assert_close(t, ref["full"], atol=atol, rtol=rtol, msg=lambda m: f"{what}: {m}")
        return
    assert list(t.shape) == ref["shape"], (what, t.shape, ref["shape"])
    st = ref.get("stride", 3)
    torch.testing.assert_close(t[..., ::st, ::st, ::st], ref["strided"], atol=atol, rtol=rtol, msg=lambda m: f"{what}: {m}")
    s = float(t.double().abs().sum())
    assert abs(s - ref["abs_sum"]) <= rtol * 10 * ref["abs_sum"] + atol, (what, s, ref["abs_sum"])


BASE = dict(n_classes=2, base_ch=16, conv_depths=[1, 1, 1, 1], kernel_sizes=[1, 3, 5],
            min_dim_group=[4, 8, 8, 16], conv_expansion_factor=[3, 3, 2, 2], attn_base_ch=16,
            depths=[1, 1, 1, 1], min_small_window_sizes=[[1, 1, 1]] * 4, min_dim_head=[4, 8, 8, 16],
            ffn_expansion_ratio=[3, 3, 2, 2], num_heads=[1, 2, 2, 4], proj_drop=0.0, conv_drop=0.0,
            attn_drop=0.0, spatial_dim=3)

CASES = {
    # G1: same 24/12/6/3 pyramid + [3,6,3,3] windows as production 96^3, at 1/8 the voxels (patch_size 2)
    "g1_48_m2": (dict(BASE, input_size=[48, 48, 48], patch_size=2, in_ch=[1, 1],
                      min_big_window_sizes=[[3] * 3, [6] * 3, [3] * 3, [3] * 3]), 2),
    # G2: BASELINE.json configs[0] shape (1,2,32,32,32); runnable only with patch_size 2 + [2,4,2,2] windows
    "g2_32_m2": (dict(BASE, input_size=[32, 32, 32], patch_size=2, in_ch=[1, 1],
                      min_big_window_sizes=[[2] * 3, [4] * 3, [2] * 3, [2] * 3]), 1),
    # G3: patch 4, 64^3, BraTS-like: one 4-channel modality, 4 classes, two PWA blocks at level 1
    "g3_64_brats": (dict(BASE, input_size=[64, 64, 64], patch_size=4, in_ch=[4], n_classes=4, depths=[2, 1, 1, 1],
                         min_big_window_sizes=[[2] * 3, [4] * 3, [2] * 3, [2] * 3]), 1),
    # G4: anisotropic Hecktor-style windows (config/models_config_hecktor2022.json), 64x64x32, patch 2
    "g4_aniso_m2": (dict(BASE, input_size=[64, 64, 32], patch_size=2, in_ch=[1, 1],
                         min_big_window_sizes=[[4, 4, 2], [8, 8, 4], [4, 4, 2], [4, 4, 2]]), 2),
    # G5 / G6: the HEADLINE shapes (BASELINE.json configs[2] / configs[1]): 128^3 patches, patch 4, windows [4,8,4,4], batch 1
    "g5_128_m2": (dict(BASE, input_size=[128, 128, 128], patch_size=4, in_ch=[1, 1],
                       min_big_window_sizes=[[4] * 3, [8] * 3, [4] * 3, [4] * 3]), 1),
    "g6_128_brats": (dict(BASE, input_size=[128, 128, 128], patch_size=4, in_ch=[4], n_classes=4,
                          min_big_window_sizes=[[4] * 3, [8] * 3, [4] * 3, [4] * 3]), 1),
    # G7: the SHIPPED AutoPET-II configuration exactly (config/models_config_autopetii.json "VeloxSeg": 96^3, patch 4, windows [3,6,3,3]), batch 1
    "g7_96_m2": (dict(BASE, input_size=[96, 96, 96], patch_size=4, in_ch=[1, 1],
                      min_big_window_sizes=[[3] * 3, [6] * 3, [3] * 3, [3] * 3]), 1),
}
BIG_CASES = ("g5_128_m2", "g6_128_brats", "g7_96_m2")      # fixtures store the arg-max mask bit-packed and coarser subsamples
LOSS_CFG = {"deep_Loss_weight": [1, 1, 1, 1], "RC_Loss_weight": 0.5, "Feature_Loss_weight": 2.0}
