#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference).  The reference imports MONAI 1.5.0,
which is not installed here; this script writes a ~40-line stand-in for the five MONAI symbols the
VeloxSeg path uses into a temporary directory and puts it on sys.path.  The
stand-in is this repo's own code and states the MONAI semantics we assumed.  Nothing of the
reference (source, bytecode) is written to the repo; only numbers (inputs, expected outputs,
seeded state-dict checksums) are.

Fixtures (all fp32, CPU, torch.manual_seed-ed; see CASES below):
  <case>.pt : dict(config, seed, state_dict, x, labels, eval_logits, argmax(uint8),
                   train_outputs (list, dropout p=0), loss, grad_norms{param->float},
                   grads_sample{param->tensor for a few small params})
  ops.pt    : per-op micro goldens (LayerNorm, gather/scatter, attention w/ bias M=2, PatchMerging,
              JLC, DownConv, UpConv, PixelShuffle, Gram, DiceLoss/CE, full Loss)
Usage: python tests/golden/make_golden.py [case ...]      (no argument = every case + ops.pt)
"""
import os
import sys
import tempfile
import types
import hashlib

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

MONAI_STUB = r'''
"""Stand-in for the 5 MONAI 1.5.0 symbols VeloxSeg's hot path imports (assumed semantics, A6)."""
import torch, torch.nn as nn, torch.nn.functional as F

class PatchEmbed(nn.Module):
    # monai.networks.blocks.PatchEmbed: pad to a multiple of p, proj = Conv(in, embed, k=p, s=p), optional norm
    def __init__(self, patch_size=2, in_chans=1, embed_dim=48, norm_layer=None, spatial_dims=3):
        super().__init__()
        p = (patch_size,) * spatial_dims if isinstance(patch_size, int) else tuple(patch_size)
        self.patch_size = p
        conv = nn.Conv3d if spatial_dims == 3 else nn.Conv2d
        self.proj = conv(in_chans, embed_dim, kernel_size=p, stride=p)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None
    def forward(self, x):
        sp = x.shape[2:]
        pads = []
        for s, p in zip(reversed(sp), reversed(self.patch_size)):
            pads += [0, (p - s % p) % p]
        if any(pads):
            x = F.pad(x, pads)
        x = self.proj(x)
        assert self.norm is None
        return x

class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__(); self.drop_prob = drop_prob
    def forward(self, x):
        assert self.drop_prob == 0.0 or not self.training
        return x

def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean, std, a, b)

def get_act_layer(name):
    assert str(name).upper() == "GELU"
    return nn.GELU()

class DiceLoss(nn.Module):
    # monai.losses.DiceLoss(include_background=False, to_onehot_y=True, softmax=True), reduction="mean",
    # smooth_nr = smooth_dr = 1e-5, batch=False, squared_pred=False, jaccard=False
    def __init__(self, include_background=True, to_onehot_y=False, softmax=False):
        super().__init__()
        self.include_background, self.to_onehot_y, self.softmax = include_background, to_onehot_y, softmax
    def forward(self, input, target):
        n = input.shape[1]
        if self.softmax:
            input = torch.softmax(input, 1)
        if self.to_onehot_y:
            target = F.one_hot(target.squeeze(1).long(), n).movedim(-1, 1).to(input.dtype)
        if not self.include_background:
            input, target = input[:, 1:], target[:, 1:]
        axes = list(range(2, input.ndim))
        inter = (input * target).sum(axes)
        den = input.sum(axes) + target.sum(axes)
        return (1.0 - (2.0 * inter + 1e-5) / (den + 1e-5)).mean()
'''


def install_stub():
    d = tempfile.mkdtemp(prefix="monai_stub_")
    for pkg in ["monai", "monai/networks", "monai/networks/blocks", "monai/networks/layers", "monai/losses"]:
        os.makedirs(os.path.join(d, pkg), exist_ok=True)
    open(os.path.join(d, "monai", "_impl.py"), "w").write(MONAI_STUB)
    open(os.path.join(d, "monai", "__init__.py"), "w").write("from . import networks, losses\n")
    open(os.path.join(d, "monai/networks/__init__.py"), "w").write("from . import blocks, layers\n")
    open(os.path.join(d, "monai/networks/blocks/__init__.py"), "w").write("from monai._impl import PatchEmbed\n")
    open(os.path.join(d, "monai/networks/layers/__init__.py"), "w").write(
        "from monai._impl import DropPath, trunc_normal_, get_act_layer\n")
    open(os.path.join(d, "monai/losses/__init__.py"), "w").write("from monai._impl import DiceLoss\n")
    sys.path.insert(0, d)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


sys.path.insert(0, HERE)
from recipe import BIG_CASES, CASES, LOSS_CFG, fill_state_dict, make_inputs, tensor_sha, sd_sha, compact, pack_mask  # noqa: E402


def make_case(name, cfg, B):
    from model.VeloxSeg import VeloxSeg
    from utils.loss import Loss
    torch.manual_seed(12345)
    model = VeloxSeg(**cfg)
    init_sha = sd_sha(model.state_dict())      # seed-level init parity (initialization.py:3-14, attention_utils.py:118)
    sd = fill_state_dict(model.state_dict(), seed=7)
    model.load_state_dict(sd)
    x, labels = make_inputs(cfg, B)
    model.eval()
    with torch.no_grad():
        logits = model(x)
    model.train()  # all dropout p = 0 in these configs
    # Work-around for a PyTorch 2.10 *CPU* autograd bug met while generating these vectors: the einsum in
    # get_pram_matrix (common_function.py:14) hands a channels-last-strided grad to the JLC backward, and a CPU
    # backward kernel then returns wrong values (finite differences + fp64 confirm; a .contiguous() on the SAME
    # gradient fixes it).  The hook below only makes that gradient contiguous: identity in exact arithmetic, the
    # reference's maths is unchanged.
    def _contig_grad(mod, inp, out):
        if out.requires_grad:
            out.register_hook(lambda g: g.contiguous())
    for dec in [model.decoder] + list(model.rc_decoders):
        dec.layer1.register_forward_hook(_contig_grad)
    outs = model(x)
    args = types.SimpleNamespace(model_name="VeloxSeg")
    crit = Loss(args, LOSS_CFG, torch.device("cpu"), num_modal=len(cfg["in_ch"]))
    loss = crit(outs, labels, sr_labels=x)
    loss.backward()
    grad_norms = {n: float(p.grad.double().norm()) for n, p in model.named_parameters()}
    small = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.numel() <= 600}
    fix = dict(config=cfg, batch=B, init_seed=12345, init_sha256=init_sha, sd_seed=7, sd_sha256=sd_sha(sd), sd_keys=list(sd.keys()),
               sd_shapes={k: list(v.shape) for k, v in sd.items()}, x_sha256=tensor_sha(x), labels_sha256=tensor_sha(labels),
               eval_logits=compact(logits), argmax=(pack_mask(logits.argmax(1).to(torch.uint8), cfg["n_classes"]) if name in BIG_CASES else logits.argmax(1).to(torch.uint8)),
               train_outputs=[compact(o) for o in outs], loss=float(loss),
               loss_cfg=LOSS_CFG, grad_norms=grad_norms, grads_small=small)
    torch.save(fix, os.path.join(HERE, name + ".pt"))
    print(name, "params", sum(p.numel() for p in model.parameters()), "loss", float(loss),
          "size MB", os.path.getsize(os.path.join(HERE, name + ".pt")) / 1e6)


def make_ops():
    """Per-op micro goldens straight from the reference's classes."""
    from model.components.attention_utils import LayerNorm, PatchMerging, FFN, PositionalEmbedding
    from model.components.PWA import MultiModal_Paired_Windows_Attention, Paired_Windows_TransformerBlock
    from model.components.conv_blocks import JLC, DownConv, UpConv
    from model.components.superpixel import PixelShuffle
    from model.components.common_function import get_pram_matrix
    from utils.loss import Loss
    from utils.runtime import normalized_deep_loss_weights, veloxseg_output_layout
    import monai
    g = torch.Generator().manual_seed(99)
    R = lambda *s: torch.randn(*s, generator=g)
    out = {}
    torch.manual_seed(5)

    def grads(y, *ts):
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
        gs = torch.autograd.grad(y, ts, gy)
        return gy, [t.detach().clone() for t in gs]

    # LayerNorm channels-first
    ln = LayerNorm(16, data_format="channels_first", dim=3)
    with torch.no_grad():
        ln.weight.add_(0.2 * R(16)); ln.bias.add_(0.2 * R(16))
    x = R(2, 16, 4, 5, 6).requires_grad_()
    y = ln(x); gy, (gx, gw, gb) = grads(y, x, ln.weight, ln.bias)
    out["layernorm"] = dict(x=x.detach(), w=ln.weight.detach().clone(), b=ln.bias.detach().clone(), y=y.detach(), gy=gy, gx=gx, gw=gw, gb=gb)

    # PWA attention module (M=2) on (1,32,12,12,12) windows [6]: 2 scales, 2 heads; incl. gather/scatter
    pwa = MultiModal_Paired_Windows_Attention(input_size=[12, 12, 12], in_channels=[32, 32], min_big_window_size=[6, 6, 6],
                                              min_small_window_size=[1, 1, 1], num_heads=2, min_dim_head=8,
                                              attn_drop=0.0, proj_drop=0.0, norm_layer=LayerNorm, dim=3)
    with torch.no_grad():
        pwa.position_embedding.relative_position_bias_table.add_(0.5 * R(*pwa.position_embedding.relative_position_bias_table.shape))
    xs = [R(1, 32, 12, 12, 12).requires_grad_() for _ in range(2)]
    q = R(1, pwa.channels_qk, 12, 12, 12)
    tok, Ns, n = pwa.window_gathering(q)
    sc = pwa.window_scattering(tok, Ns, n)
    ys = pwa(xs)
    gy = [torch.randn(y.shape, generator=torch.Generator().manual_seed(3 + i)) for i, y in enumerate(ys)]
    params = dict(pwa.named_parameters())
    gs = torch.autograd.grad(ys, xs + list(params.values()), gy)
    out["pwa"] = dict(sd={k: v.detach().clone() for k, v in pwa.state_dict().items()}, xs=[t.detach() for t in xs],
                      q=q, tok=tok, Ns=Ns, n=n, scat=sc, ys=[t.detach() for t in ys], gy=gy,
                      gxs=[t.clone() for t in gs[:2]], gparams={k: t.clone() for k, t in zip(params, gs[2:])},
                      channels_qk=pwa.channels_qk, channels_v=pwa.channels_v,
                      big=pwa.big_window_size, small=pwa.small_window_size)

    # Transformer block (double residual + FFN), M=1, L1-like: C=16, grid 8, windows 2 -> scales 2,4,8
    blk = Paired_Windows_TransformerBlock(input_size=[8, 8, 8], in_channels=[16], min_big_window_size=[2, 2, 2],
                                          num_heads=1, min_dim_head=4, attn_drop=0.0, proj_drop=0.0, drop_path=0.0,
                                          ffn_expansion_ratio=3, dim=3)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.add_(0.05 * R(*p_.shape))
    x = R(2, 16, 8, 8, 8)
    out["block"] = dict(sd={k: v.detach().clone() for k, v in blk.state_dict().items()}, x=x, y=blk([x])[0].detach(),
                        channels_qk=blk.attn.channels_qk, channels_v=blk.attn.channels_v)

    # PatchMerging
    pm = PatchMerging(16, norm_layer=LayerNorm, dim=3)
    with torch.no_grad():
        pm.norm.weight.add_(0.2 * R(128)); pm.norm.bias.add_(0.2 * R(128))
    x = R(2, 16, 4, 6, 8).requires_grad_()
    y = pm(x); gy, gs = grads(y, x, *pm.parameters())
    out["patchmerge"] = dict(sd={k: v.detach().clone() for k, v in pm.state_dict().items()}, x=x.detach(), y=y.detach(), gy=gy,
                             gx=gs[0], gparams={k: t for k, t in zip(dict(pm.named_parameters()), gs[1:])})

    # JLC (1,16,8,8,8) groups 4, expansion 3
    jlc = JLC(16, kernel_sizes=[1, 3, 5], groups=4, epansion_factor=3, dropout=0.0, spatial_dim=3)
    with torch.no_grad():
        for p_ in jlc.parameters():
            if p_.ndim == 1:
                p_.add_(0.1 * R(*p_.shape))
    x = R(2, 16, 8, 8, 8).requires_grad_()
    y = jlc(x); gy, gs = grads(y, x, *jlc.parameters())
    out["jlc"] = dict(sd={k: v.detach().clone() for k, v in jlc.state_dict().items()}, x=x.detach(), y=y.detach(), gy=gy,
                      gx=gs[0], gparams={k: t for k, t in zip(dict(jlc.named_parameters()), gs[1:])})

    # DownConv patch 4 (k7 s4 p3) and patch 2 (k3 s2 p1); UpConv
    for nm, mod, x in [("down4", DownConv(2, 16, patch_size=4), R(2, 2, 16, 16, 16)),
                       ("down2", DownConv(16, 32, patch_size=2), R(2, 16, 8, 8, 8)),
                       ("up2", UpConv(32, 16, up_rate=2), R(2, 32, 4, 4, 4))]:
        with torch.no_grad():
            for p_ in mod.parameters():
                if p_.ndim == 1:
                    p_.add_(0.1 * R(*p_.shape))
        x = x.requires_grad_()
        y = mod(x); gy, gs = grads(y, x, *mod.parameters())
        out[nm] = dict(sd={k: v.detach().clone() for k, v in mod.state_dict().items()}, x=x.detach(), y=y.detach(), gy=gy,
                       gx=gs[0], gparams={k: t for k, t in zip(dict(mod.named_parameters()), gs[1:])})

    # PixelShuffle, Gram
    x = R(2, 128, 3, 4, 5)
    out["pixelshuffle"] = dict(x=x, y=PixelShuffle(4, 3)(x))
    x = R(2, 16, 6, 6, 6)
    out["gram"] = dict(x=x, y=get_pram_matrix(x))

    # Dice / CE / full Loss on (2, ncls, 16^3)
    for ncls in (2, 4):
        logit = R(2, ncls, 16, 16, 16).requires_grad_()
        lab = torch.randint(0, ncls, (2, 1, 16, 16, 16), generator=g)
        dl = monai.losses.DiceLoss(include_background=False, to_onehot_y=True, softmax=True)(logit, lab)
        ce = torch.nn.CrossEntropyLoss()(logit, lab.squeeze(1))
        gl, = torch.autograd.grad(dl + ce, logit)
        out[f"segloss{ncls}"] = dict(logit=logit.detach(), lab=lab.to(torch.uint8), dice=float(dl.detach()), ce=float(ce.detach()), glogit=gl)
    args = types.SimpleNamespace(model_name="VeloxSeg")
    lcfg = {"deep_Loss_weight": [1, 1, 1, 1], "RC_Loss_weight": 0.5, "Feature_Loss_weight": 2.0}
    crit = Loss(args, lcfg, torch.device("cpu"), num_modal=2)
    outs = [R(2, 2, 8, 8, 8).requires_grad_() for _ in range(4)] + [R(2, 2, 8, 8, 8).requires_grad_()] + \
           [R(2, 16, 16).requires_grad_() for _ in range(3)]
    lab = torch.randint(0, 2, (2, 1, 8, 8, 8), generator=g)
    sr = R(2, 2, 8, 8, 8)
    L = crit(outs, lab, sr_labels=sr)
    gs = torch.autograd.grad(L, outs)
    out["loss_full"] = dict(outs=[o.detach() for o in outs], lab=lab.to(torch.uint8), sr=sr, loss=float(L), gouts=list(gs), cfg=lcfg)
    # runtime helper known answers (reference tests/test_runtime_helpers.py:63-75,87-111)
    out["runtime"] = dict(w5=normalized_deep_loss_weights([1, 1, 1, 1], 5), w4=normalized_deep_loss_weights([4, 2, 1, 1], 4),
                          layout8=veloxseg_output_layout(8, 2), layout5=veloxseg_output_layout(5, 2))
    torch.save(out, os.path.join(HERE, "ops.pt"))
    print("ops.pt MB", os.path.getsize(os.path.join(HERE, "ops.pt")) / 1e6)


if __name__ == "__main__":
    import atexit, shutil
    install_stub()
    atexit.register(lambda: shutil.rmtree(sys.path[1], ignore_errors=True) if "monai_stub_" in sys.path[1] else None)
    torch.set_num_threads(8)
    only = sys.argv[1:]                      # optional: case names to (re)generate; default = everything
    if not only:
        make_ops()
    for name, (cfg, B) in CASES.items():
        if not only or name in only:
            make_case(name, cfg, B)
