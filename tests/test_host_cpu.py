"""CPU-side checks: module tree / state-dict contract, seed-level init parity with the reference, host geometry,
runtime helpers, the C-ABI library (loads, exports every declared symbol; no compute without a GPU)."""
import ctypes
import os

import pytest
import torch

from oracle import veloxseg_oracle as O
from recipe import CASES, sd_sha


def test_library_exports_every_header_symbol():
    import __graft_entry__ as G
    G.build()
    from veloxseg_amd import _hip
    protos = _hip.parse_header()
    assert len(protos) >= 30 and "vx_conv3d_fwd" in protos and "vx_pwa_attn_bwd" in protos
    dll = ctypes.CDLL(_hip.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/veloxseg_hip.h but not exported"
    assert dll.vx_abi_version() == _hip.ABI_VERSION
    assert ctypes.sizeof(_hip.VxPwaPlan) == 4 * (3 + 3 + 2 + 12 + 12 + 4 + 2)


def test_bad_arguments_return_error_codes_without_gpu():
    from veloxseg_amd import _hip
    with pytest.raises(RuntimeError, match="bad sizes|null pointer"):
        _hip.call("vx_conv3d_fwd", None, None, 0, None, None, None, 0, 4, 4, 4, 4, 4, 1, 1, 0, 1, 1, None)
    with pytest.raises(RuntimeError, match="not divisible"):
        _hip.call("vx_conv3d_fwd", 1, None, 0, 1, None, 1, 1, 6, 4, 4, 4, 4, 1, 1, 0, 4, 1, None)


@pytest.mark.parametrize("name", list(CASES))
def test_state_dict_contract_and_seed_init(golden_dir, name):
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    cfg_d, _ = CASES[name]
    fix = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    torch.manual_seed(fix["init_seed"])
    model = VeloxSeg(**cfg_d)
    sd = model.state_dict()
    assert list(sd.keys()) == fix["sd_keys"], "state_dict keys/order must equal the reference's"
    for k, v in sd.items():
        assert list(v.shape) == fix["sd_shapes"][k], k
    assert sd_sha(sd) == fix["init_sha256"], "same seed must give the reference's initial weights (He init + trunc_normal tables)"
    tmpl = O.state_dict_template(O.OracleConfig(**cfg_d))
    assert list(tmpl.keys()) == list(sd.keys())
    for k in sd:
        if not torch.is_floating_point(sd[k]):
            assert torch.equal(sd[k], tmpl[k]), k     # relative_position_index buffers


def test_shipped_config_kwargs_are_accepted():
    """config/models_config_*.json 'VeloxSeg' entries (values restated here; the JSON files are not read at test time)."""
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    base = {"patch_size": 4, "base_ch": 16, "conv_depths": [1, 1, 1, 1], "kernel_sizes": [1, 3, 5], "min_dim_group": [4, 8, 8, 16],
            "conv_expansion_factor": [3, 3, 2, 2], "attn_base_ch": 16, "depths": [1, 1, 1, 1], "min_small_window_sizes": [[1, 1, 1]] * 4,
            "min_dim_head": [4, 8, 8, 16], "ffn_expansion_ratio": [3, 3, 2, 2], "proj_drop": 0.1, "conv_drop": 0.1, "spatial_dim": 3}
    autopet = dict(base, input_size=[96, 96, 96], in_ch=[1, 1], n_classes=2, num_heads=[1, 2, 2, 4], min_big_window_sizes=[[3] * 3, [6] * 3, [3] * 3, [3] * 3])
    brats = dict(base, input_size=[96, 96, 96], in_ch=[4], n_classes=4, num_heads=[1, 2, 2, 4], min_big_window_sizes=[[3] * 3, [6] * 3, [3] * 3, [3] * 3])
    hecktor = dict(base, input_size=[128, 128, 64], in_ch=[1, 1], n_classes=2, min_big_window_sizes=[[4, 4, 2], [8, 8, 4], [4, 4, 2], [4, 4, 2]])
    n = {}
    for nm, cfg in (("autopet", autopet), ("brats", brats), ("hecktor", hecktor)):
        m = VeloxSeg(**cfg)
        n[nm] = sum(p.numel() for p in m.parameters())
        assert m.encoder.encoder_attn.layers[0].blocks[0].attn.attn_drop == 0.1      # ctor default (VeloxSeg.py:83)
    assert n == {"autopet": 2288999, "brats": 1863117, "hecktor": 2289641}, n       # SURVEY / BASELINE.md parameter counts


def test_window_plan_matches_oracle_and_rejects_bad_tiling():
    from veloxseg_amd.model.components.PWA import plan_windows
    for grid, big, heads, mdh, C in ([24] * 3, [3] * 3, 1, 4, 16), ([16] * 3, [8] * 3, 2, 8, 32), ([32, 32, 16], [4, 4, 2], 1, 4, 16), ([3] * 3, [3] * 3, 4, 16, 128):
        a = plan_windows(grid, big, [1, 1, 1], 2, heads, mdh, C)
        b = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
        assert (a["big"], a["small"], a["n"], a["nwin"], a["ch_qk"], a["ch_v"]) == (b["big"], b["small"], b["n"], b["nwin"], b["ch_qk"], b["ch_v"])
    assert len(plan_windows([24] * 3, [3] * 3, [1] * 3, 2, 1, 4, 16)["big"]) == 4          # 3,6,12,24
    with pytest.raises(ValueError, match="tile"):
        plan_windows([32] * 3, [3] * 3, [1] * 3, 2, 1, 4, 16)                                # [3,6,3,3] does not divide 32
    with pytest.raises(ValueError, match="tile"):
        plan_windows([32, 32, 16], [4, 4, 4], [1] * 3, 2, 1, 4, 16)                          # `.any()` admits a scale that cannot tile axis 2


def test_runtime_helpers_known_answers():
    """reference tests/test_runtime_helpers.py:36-42,63-75,87-111 against the product helpers."""
    from veloxseg_amd.utils.runtime import expected_input_channels, normalized_deep_loss_weights, veloxseg_output_layout
    assert normalized_deep_loss_weights([1, 1, 1, 1], 5) == [0.2] * 5
    with pytest.raises(ValueError, match="deep_Loss_weight"):
        normalized_deep_loss_weights([4, 2, 1], 5)
    with pytest.raises(ValueError, match="sum"):
        normalized_deep_loss_weights([0, 0, 0, 0], 5)
    assert veloxseg_output_layout(output_count=8, num_modal=2) == {"seg": (0, 4), "reconstruction": 4, "decoder_gram": 5, "teacher_grams": (6, 7)}
    assert veloxseg_output_layout(output_count=5, num_modal=2) == {"seg": (0, 1), "reconstruction": 1, "decoder_gram": 2, "teacher_grams": (3, 4)}
    with pytest.raises(ValueError, match="VeloxSeg"):
        veloxseg_output_layout(output_count=4, num_modal=2)
    assert expected_input_channels("VeloxSeg", {"VeloxSeg": {"in_ch": [1, 1]}}) == 2


def test_no_cpu_fallback_and_no_oracle_import_in_product():
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    cfg_d, _ = CASES["g2_32_m2"]
    with pytest.raises(RuntimeError, match="MI355X"):
        VeloxSeg(**cfg_d)(torch.zeros(1, 2, 32, 32, 32))
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "veloxseg_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} imports the oracle"


def test_fastcall_extension_wraps_every_entry_point():
    """the generated CPython wrappers (csrc/_vxfast.c) cover every int-returning prototype of the header and sit on the same library"""
    from veloxseg_amd import _hip as H
    fast = H.fast_module()
    assert fast is not None, "run `python -c 'import __graft_entry__ as g; g.build()'`"
    protos = H.parse_header()
    missing = [n for n, (ret, _) in protos.items() if not ret.startswith("const char") and not hasattr(fast, n)]
    assert not missing, missing
    assert fast.vx_abi_version() == H.ABI_VERSION
    import pytest
    with pytest.raises(TypeError):
        fast.vx_add(0, 0, 0)                          # wrong arity: nothing is launched
    with pytest.raises(TypeError):
        fast.vx_pwa_attn_set_split("two")             # wrong type: nothing is launched
    assert fast.vx_pwa_attn_set_split(3) != 0 and b"must be" in H.LIB.load().vx_last_error()
    assert fast.vx_pwa_attn_set_split(0) == 0


def test_cpp_operator_module_builds_and_loads():
    """veloxseg_amd._vxops (C++ operator bodies) imports on a CPU-only box and refuses CPU tensors like the python bodies do"""
    import pytest
    import torch
    from veloxseg_amd import functional as VF
    m = VF.cpp_module()
    assert m is not None, "run `python -c 'import __graft_entry__ as g; g.build()'`"
    for name in ("conv_fwd", "conv_bwd", "in_fwd", "in_bwd", "ln_fwd", "ln_bwd", "gelu_fwd", "gelu_bwd", "axpy_fwd", "axpy_bwd", "jlc_fwd", "jlc_bwd", "ffn_fwd", "ffn_bwd"):
        assert hasattr(m, name), name
    with pytest.raises(RuntimeError, match="no CPU"):
        VF.conv3d(torch.zeros(1, 4, 4, 4, 4), torch.zeros(4, 4, 1, 1, 1))
