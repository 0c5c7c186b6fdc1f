"""The hot-path operators as registered PyTorch custom ops: torch.ops.veloxseg.* (north_star: "re-registered as custom ops backed by hand-written
HIP kernels").  Each op is the operator of veloxseg_amd.functional under a dispatcher schema; the implementation is registered for
the Autograd / CUDA / Meta keys from C++ for the seven pure operators (csrc/_vxops.cpp) and for CompositeImplicitAutograd for the four that need Python-side
state; autograd records the operator's own node (a C++ or Python autograd function whose forward and backward launch the kernels of include/veloxseg_hip.h), and
a CPU tensor raises the library's "no CPU path" error -- there is no fallback kernel behind any key.

    torch.ops.veloxseg.conv3d(x, w, b, stride, padding, groups, pixel_shuffle)        nn.Conv3d (+ PixelShuffle), conv_blocks.py / Decoder.py
    torch.ops.veloxseg.conv_transpose_k2s2(x, w, b)                                   nn.ConvTranspose3d(k=2, s=2), conv_blocks.py:29-35
    torch.ops.veloxseg.instance_norm_sum(ys, act, res)                                sum_k [GELU](InstanceNorm3d(y_k)) [+ res]
    torch.ops.veloxseg.layer_norm_cf(x, gamma, beta)                                  channels-first LayerNorm, attention_utils.py:29-43
    torch.ops.veloxseg.gelu_dropout(a, p, site) / residual_dropout(x, z, alpha, p, site)
    torch.ops.veloxseg.space_to_depth2(x) / upsample_trilinear(x, size) / gram(x)
    torch.ops.veloxseg.pwa_attention(table, qkv, grid, n, heads, small, nwin, cq, cv, p_attn, site, rng_state=None)     PWA.py:106-200,308-327
    torch.ops.veloxseg.jlc_block(x, ws, bs, groups, l1w, l1b, l2w, l2b, p, site, rng_state=None)                         conv_blocks.py:41-75
    torch.ops.veloxseg.ffn_tail(y, gamma, beta, w1, b1, w2, b2, p, site1, site2, rng_state=None)                         attention_utils.py:45-71 (LN + FFN + residual)
    torch.ops.veloxseg.seg_loss(outputs, labels, sr_labels, head_weights, w_rc, w_f, num_modal)                          utils/loss.py:52-66

The modules of veloxseg_amd.model reach the seven C++-registered operators THROUGH the dispatcher (functional.py routes conv3d, conv_transpose_k2s2, instance_norm_sum,
layer_norm_cf, space_to_depth2, upsample_trilinear and gram to torch.ops.veloxseg.*); the composite operators are called as functions of veloxseg_amd.functional.
With VELOXSEG_NO_CPP=1 (the documented A/B mode without the C++ module) all eleven schemas are defined here under CompositeImplicitAutograd, as before round 4."""
import os
from typing import List, Optional

import torch

from . import _hip as H
from . import functional as VF

# Eleven operators are DEFINED AND REGISTERED IN C++ (csrc/_vxops.cpp: TORCH_LIBRARY(veloxseg) with Autograd / CUDA / Meta keys and a CPU key that raises): the seven
# generic ones (conv3d, conv_transpose_k2s2, instance_norm_sum, layer_norm_cf, space_to_depth2, upsample_trilinear, gram) and, since round 5, the four north_star names --
# pwa_attention (the window plan as integer lists, the dropout site and the {seed, step} RNG-state tensor as arguments), jlc_block, ffn_tail, seg_loss.  Importing the
# extension registers them.  gelu_dropout / residual_dropout stay Python bodies under CompositeImplicitAutograd (they fetch the RNG-state tensor themselves).
CPP_OPS = ("conv3d", "conv_transpose_k2s2", "instance_norm_sum", "layer_norm_cf", "space_to_depth2", "upsample_trilinear", "gram",
           "pwa_attention", "jlc_block", "ffn_tail", "seg_loss")          # (round 5: the four operators north_star names are C++ dispatcher ops too)
_CPP = VF.cpp_module()
if _CPP is None and os.environ.get("VELOXSEG_NO_CPP") != "1":
    raise RuntimeError("veloxseg_amd.ops: the C++ operator module (veloxseg_amd._vxops) is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(VELOXSEG_NO_CPP=1 selects the Python operator bodies on purpose)")
_lib = torch.library.Library("veloxseg", "FRAGMENT")
_plans = {}


def _define(schema, fn):
    if _CPP is not None and schema.split("(")[0] in CPP_OPS:
        return                              # (kept in the list below as documentation of the schema: C++ owns it)
    _lib.define(schema)
    _lib.impl(schema.split("(")[0], fn, "CompositeImplicitAutograd")


def _conv3d(x, w, b, stride, padding, groups, pixel_shuffle):
    return VF.conv3d(x, w, b, stride=stride, padding=padding, groups=groups, pixel_shuffle=pixel_shuffle)


def _plan(grid, n, heads, small, nwin):
    key = (tuple(grid), tuple(n), int(heads), tuple(small), tuple(nwin))
    if key not in _plans:
        nb = len(small) // 3
        _plans[key] = H.make_plan(list(grid), list(n), int(heads), [list(small[3 * i:3 * i + 3]) for i in range(nb)], [list(nwin[3 * i:3 * i + 3]) for i in range(nb)])
    return _plans[key]


def _pwa_attention(table, qkv: List[torch.Tensor], grid: List[int], n: List[int], heads: int, small: List[int], nwin: List[int], cq: int, cv: int,
                   p_attn: float, site: int):
    return VF.pwa_core(table, _plan(grid, n, heads, small, nwin), cq, cv, list(qkv), p_attn=p_attn, site=site)


def _seg_loss(outputs: List[torch.Tensor], labels, sr_labels: Optional[torch.Tensor], head_weights: List[float], w_rc: float, w_f: float, num_modal: int):
    if num_modal <= 0:
        return VF.seg_only_loss(list(outputs), labels, list(head_weights))
    return VF.veloxseg_loss(list(outputs), labels, sr_labels, list(head_weights), w_rc, w_f, num_modal)


_define("conv3d(Tensor x, Tensor w, Tensor? b, int stride, int padding, int groups, int pixel_shuffle) -> Tensor", _conv3d)
_define("conv_transpose_k2s2(Tensor x, Tensor w, Tensor b) -> Tensor", lambda x, w, b: VF.conv_transpose_k2s2(x, w, b))
_define("instance_norm_sum(Tensor[] ys, bool act, Tensor? res) -> Tensor", lambda ys, act, res: VF.instnorm_sum(list(ys), act=act, res=res))
_define("layer_norm_cf(Tensor x, Tensor gamma, Tensor beta) -> Tensor", lambda x, g, b: VF.layernorm_cf(x, g, b))
_define("gelu_dropout(Tensor a, float p, int site) -> Tensor", lambda a, p, site: VF.gelu_dropout(a, p, site))
_define("residual_dropout(Tensor? x, Tensor z, float alpha, float p, int site) -> Tensor", lambda x, z, alpha, p, site: VF.residual_dropout(x, z, alpha, p, site))
_define("space_to_depth2(Tensor x) -> Tensor", lambda x: VF.space_to_depth2(x))
_define("upsample_trilinear(Tensor x, int[] size) -> Tensor", lambda x, size: VF.upsample_trilinear(x, tuple(size)))
_define("gram(Tensor x) -> Tensor", lambda x: VF.gram(x))
_define("pwa_attention(Tensor table, Tensor[] qkv, int[] grid, int[] n, int heads, int[] small, int[] nwin, int cq, int cv, float p_attn, int site) -> Tensor[]",
        _pwa_attention)
_define("seg_loss(Tensor[] outputs, Tensor labels, Tensor? sr_labels, float[] head_weights, float w_rc, float w_f, int num_modal) -> Tensor", _seg_loss)

OPS = ("conv3d", "conv_transpose_k2s2", "instance_norm_sum", "layer_norm_cf", "gelu_dropout", "residual_dropout", "space_to_depth2", "upsample_trilinear", "gram",
       "pwa_attention", "seg_loss") + (("jlc_block", "ffn_tail") if _CPP is not None else ())
