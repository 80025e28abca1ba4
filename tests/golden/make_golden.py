#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run only in the build container (needs /root/reference):

    python tests/golden/make_golden.py

The reference's Python is imported unmodified through the import shim of
SURVEY.md 8c (an empty stub package `models` whose __path__ points at the
reference's models/ directory, so models/__init__.py -- which needs torchvision --
is never executed).  Nothing from the reference is written into the repo: the
fixtures hold tensors only (inputs, injected masks, expected outputs, expected
index lists), plus plain-python config dicts.  Parameters are produced by
tests/golden/fill.py from a seed, identically here and in the tests.
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os
import sys
import types

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from fill import fill_state_dict, seeded_bernoulli, seeded_randn  # noqa: E402

REF_ROOT = "/root/reference/imagenet_classification"


def load_reference():
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF_ROOT, "models")]
    sys.modules["models"] = pkg
    mods = {}
    for name in ("utils", "laud_resnet"):
        spec = importlib.util.spec_from_file_location(f"models.{name}", os.path.join(REF_ROOT, "models", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"models.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["utils"], mods["laud_resnet"]


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().clone().contiguous()
    if isinstance(obj, (list, tuple)):
        return [to_cpu(o) for o in obj]
    return obj


def inject(masker, mask):
    """Replace a reference masker's forward by one returning `mask` (recipe (B));
    the FLOPs figure is still the reference's own (it only depends on shapes)."""
    orig = masker.forward

    def fwd(x, temperature):
        _, _, flops = orig(x, temperature)
        return mask, mask.mean(), flops
    masker.forward = fwd
    return orig


# ----------------------------------------------------------------------------- L1
def make_l1(U):
    out = {}
    x = seeded_randn((2, 6, 3, 3), 11)
    m = torch.tensor([[1., 0., 1.], [0., 1., 1.]])
    out["chan_mask"] = dict(x=x, mask=m, y=U.apply_channel_mask(x, m))
    m6 = seeded_bernoulli((2, 6), 0.5, 12)
    out["chan_mask_full"] = dict(x=x, mask=m6, y=U.apply_channel_mask(x, m6))
    x4 = seeded_randn((2, 6, 4, 4), 13)
    for g in (1, 2, 6):
        mg = seeded_bernoulli((2, g, 4, 4), 0.5, 14 + g)
        out[f"spat_mask_g{g}"] = dict(x=x4, mask=mg, y=U.apply_spatial_mask(x4, mg))

    exp = []
    for (s, p, g) in [(1, 0, 1), (1, 1, 1), (2, 1, 1), (2, 1, 2), (1, 1, 2), (2, 0, 1)]:
        mk = seeded_bernoulli((2, g, 5, 5), 0.3, 20 + s * 7 + p * 3 + g)
        y = U.ExpandMask(stride=s, padding=p, mask_channel_group=g)(mk)
        exp.append(dict(stride=s, padding=p, groups=g, mask=mk, y=y))
    single = torch.zeros(1, 1, 3, 3)
    single[0, 0, 1, 1] = 1
    exp.append(dict(stride=2, padding=1, groups=1, mask=single,
                    y=U.ExpandMask(stride=2, padding=1)(single)))
    out["expand"] = exp

    near = []
    for (s, h) in [(3, 14), (1, 7), (7, 28), (3, 28), (5, 15), (14, 56), (7, 49), (9, 28)]:
        src = torch.arange(s * s, dtype=torch.float32).view(1, 1, s, s)
        near.append(dict(s=s, h=h, y=F.interpolate(src, size=h, mode="nearest").to(torch.int32)))
    out["nearest"] = near

    pools = []
    for (h, s) in [(28, 3), (14, 3), (56, 7), (7, 1), (28, 9)]:
        xp = seeded_randn((1, 2, h, h), 30 + h + s)
        pools.append(dict(h=h, s=s, x=xp, y=F.adaptive_avg_pool2d(xp, s)))
    out["adaptive_pool"] = pools

    maskers = []
    for (cin, g, S, hin) in [(16, 1, 4, 8), (16, 1, 8, 8), (16, 2, 4, 8), (8, 1, 3, 14), (8, 1, 1, 7)]:
        mk = quiet(U.Masker_spatial, cin, g, S).eval()
        sd = fill_state_dict(mk.state_dict(), 40 + S)
        mk.load_state_dict(sd)
        xm = F.relu(seeded_randn((3, cin, hin, hin), 41 + S))
        with torch.no_grad():
            mask, sp, fl = mk(xm, 1.0)
            pooled = F.adaptive_avg_pool2d(xm, S) if S < hin else xm
            logits = mk.conv(pooled)
        maskers.append(dict(kind="spatial", cin=cin, groups=g, mask_size=S, x=xm, sd=sd, mask=mask,
                            sparsity=sp, flops=int(fl), logits=logits))
    # exact tie: zero weights, equal biases -> keep (>=)
    mk = quiet(U.Masker_spatial, 4, 1, 2).eval()
    sd = {k: torch.zeros_like(v) for k, v in mk.state_dict().items()}
    mk.load_state_dict(sd)
    xm = seeded_randn((1, 4, 2, 2), 49)
    with torch.no_grad():
        mask, sp, fl = mk(xm, 1.0)
    maskers.append(dict(kind="spatial", cin=4, groups=1, mask_size=2, x=xm, sd=sd, mask=mask, sparsity=sp,
                        flops=int(fl), logits=torch.zeros(1, 2, 2, 2)))
    for layers in (1, 2):
        mk = U.Masker_channel_MLP(32, 8, layers=layers, reduction=16).eval()
        sd = fill_state_dict(mk.state_dict(), 50 + layers)
        mk.load_state_dict(sd)
        xm = F.relu(seeded_randn((4, 32, 6, 6), 52))
        with torch.no_grad():
            mask, sp, fl = mk(xm, 1.0)
            logits = mk.conv(F.adaptive_avg_pool2d(xm, 1).view(4, 32))
        maskers.append(dict(kind="mlp", cin=32, groups=8, layers=layers, reduction=16, x=xm, sd=sd, mask=mask,
                            sparsity=sp, flops=int(fl), logits=logits))
    mk = U.Masker_channel_conv_linear(32, 8, reduction=4).eval()
    sd = fill_state_dict(mk.state_dict(), 55)
    mk.load_state_dict(sd)
    xm = F.relu(seeded_randn((4, 32, 6, 6), 56))
    with torch.no_grad():
        mask, sp, fl = mk(xm, 1.0)
    maskers.append(dict(kind="conv_linear", cin=32, groups=8, reduction=4, x=xm, sd=sd, mask=mask,
                        sparsity=sp, flops=int(fl)))
    out["maskers"] = maskers
    return out


# ----------------------------------------------------------------------------- L2
BLOCK_CASES = {
    # name: (dyn_mode, spatial_granularity, channel_granularity, spatial_groups)
    "spatial_g1": ("spatial", 1, 1, 1),
    "spatial_g4": ("spatial", 4, 1, 1),
    "spatial_g2_grp2": ("spatial", 2, 1, 2),
    "layer": ("layer", 1, 1, 1),
    "channel_g1": ("channel", 1, 1, 1),
    "channel_g2": ("channel", 1, 2, 1),
    "both": ("both", 2, 2, 1),
}


def make_block(U, R, name, stride, seed):
    mode, sgran, cgran, sgrp = BLOCK_CASES[name]
    planes, out_size, batch = 16, 14, 3
    inplanes = 64 if stride == 1 else 32
    hin = out_size * stride
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = torch.nn.Sequential(U.conv1x1(inplanes, planes * 4, stride), torch.nn.BatchNorm2d(planes * 4))
    kw = dict(inplanes=inplanes, planes=planes, stride=stride, spatial_mask_channel_group=sgrp,
              channel_dyn_granularity=cgran, output_size=out_size, mask_spatial_granularity=sgran,
              dyn_mode=mode, channel_masker="MLP", channel_masker_layers=2, reduction=16)
    blk = quiet(R.Bottleneck, downsample=down, **kw).eval()
    sd = fill_state_dict(blk.state_dict(), seed)
    blk.load_state_dict(sd)
    x = F.relu(seeded_randn((batch, inplanes, hin, hin), seed + 1))
    def t0():  # fresh flops accumulator per run: the reference adds to it IN PLACE (laud_resnet.py:146)
        return (x, None, None, None, None, None, torch.tensor(0.0))
    # x is not stored: tests rebuild it as relu(seeded_randn(x_shape, x_seed))
    fx = dict(kw=kw, seed=seed, x_seed=seed + 1, x_shape=list(x.shape), has_downsample=down is not None)
    with torch.no_grad():
        fx["masker_run"] = to_cpu(blk(t0(), 1.0))
        if blk.masker_spatial is not None:
            m3p, _, _ = blk.masker_spatial(x, 1.0)
            fx["masker_spatial_mask"] = m3p.clone()
        if blk.masker_channel is not None:
            mc, _, _ = blk.masker_channel(x, 1.0)
            fx["masker_channel_mask"] = mc.clone()
        # injected masks
        if blk.masker_spatial is not None:
            ms = blk.mask_size
            sm = seeded_bernoulli((batch, sgrp, ms, ms), 0.5, seed + 2)
            if mode == "layer":
                sm = torch.tensor([1., 0., 1.]).view(batch, 1, 1, 1)
            inject(blk.masker_spatial, sm)
            fx["spatial_mask"] = sm
            m3 = F.interpolate(sm, size=out_size, mode="nearest")
            m2 = blk.mask_expander2(m3)
            m1 = blk.mask_expander1(m2)
            fx["mask3_px"] = m3.to(torch.bool)
            fx["mask1_px"] = m1
            if sgrp == 1:
                fx["idx3"] = torch.nonzero(m3.flatten()).flatten().to(torch.int32)
                fx["idx1"] = torch.nonzero(m1.flatten()).flatten().to(torch.int32)
        if blk.masker_channel is not None:
            g = blk.masker_channel.channel_dyn_group
            cm = seeded_bernoulli((batch, g), 0.6, seed + 3)
            cm[0, 0] = 1.0
            inject(blk.masker_channel, cm)
            fx["channel_mask"] = cm
        fx["injected_run"] = to_cpu(blk(t0(), 1.0))
    return fx


# ----------------------------------------------------------------------------- L3
FULL_CASES = {
    "r50_spatial_g1": ("uni_resnet50", dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[1, 1, 1, 1])),
    "r101_channel2222": ("uni_resnet101", dict(dyn_mode=["channel"] * 4, channel_dyn_granularity=[2, 2, 2, 2],
                                                channel_masker=["MLP"] * 4, channel_masker_layers=[2, 2, 2, 2],
                                                reduction_ratio=[16] * 4)),
    "r101_spatial4421": ("uni_resnet101", dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[4, 4, 2, 1])),
    "r50_spatial4444": ("uni_resnet50", dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[4, 4, 4, 4])),
    "r101_layer": ("uni_resnet101", dict(dyn_mode=["layer"] * 4)),
    "r50_layer_via_gran": ("uni_resnet50", dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[56, 28, 14, 7])),
    "r50_both": ("uni_resnet50", dict(dyn_mode=["both"] * 4, mask_spatial_granularity=[4, 4, 2, 1],
                                      channel_dyn_granularity=[2, 2, 2, 2], channel_masker=["MLP"] * 4,
                                      channel_masker_layers=[2, 2, 2, 2])),
    "r50_mixed": ("uni_resnet50", dict(dyn_mode=["spatial", "channel", "both", "layer"],
                                       mask_spatial_granularity=[2, 1, 2, 1], channel_dyn_granularity=[1, 4, 2, 1],
                                       channel_masker=["MLP"] * 4, channel_masker_layers=[1, 2, 1, 2])),
}


def injected_masks_for(model_blocks, batch, seed, p_spatial=0.5, p_channel=0.62):
    """Deterministic per-block Bernoulli masks; shared verbatim with the tests
    (tests/helpers.py re-implements the same few lines against the oracle blocks)."""
    masks = {}
    for i, (name, blk) in enumerate(model_blocks):
        entry = {}
        if blk.masker_spatial is not None:
            ms, g = blk.masker_spatial.mask_size, blk.masker_spatial.mask_channel_group
            entry["spatial"] = seeded_bernoulli((batch, g, ms, ms), p_spatial, seed + 2 * i)
        if blk.masker_channel is not None:
            entry["channel"] = seeded_bernoulli((batch, blk.masker_channel.channel_dyn_group), p_channel,
                                                seed + 2 * i + 1)
        masks[name] = entry
    return masks


def make_full(R, name, seed=7, batch=2, width_mult=0.125, input_size=224):
    factory, kw = FULL_CASES[name]
    kw = dict(kw, width_mult=width_mult, input_size=input_size, num_classes=1000)
    model = quiet(getattr(R, factory), **kw).eval()
    model.load_state_dict(fill_state_dict(model.state_dict(), seed))
    x = seeded_randn((batch, 3, input_size, input_size), seed + 100)
    fx = dict(factory=factory, kw=kw, seed=seed, x_seed=seed + 100, batch=batch,
              n_params=sum(p.numel() for p in model.parameters()),
              keys=list(model.state_dict().keys()))
    with torch.no_grad():
        fx["masker_run"] = to_cpu(model(x, 1.0))
        blocks = [(f"layer{s}.{j}", b) for s in (1, 2, 3, 4) for j, b in enumerate(getattr(model, f"layer{s}"))]
        masks = injected_masks_for(blocks, batch, seed=1000 + seed)
        for bname, blk in blocks:
            if "spatial" in masks[bname]:
                inject(blk.masker_spatial, masks[bname]["spatial"])
            if "channel" in masks[bname]:
                inject(blk.masker_channel, masks[bname]["channel"])
        fx["mask_seed"] = 1000 + seed
        fx["injected_run"] = to_cpu(model(x, 1.0))
    return fx


# ----------------------------------------------------------------------------- RegNet (needs a torchvision stub)
def load_reference_regnet():
    """laud_regnet.py imports 4 symbols from torchvision 0.14 (laud_regnet.py:14-16); torchvision is not installed, so a
    stub exposing exactly those is injected (SURVEY 8c shim 2).  The two containers are restated from torchvision 0.14.1."""
    import torch.nn as nn

    class ConvNormActivation(nn.Sequential):
        def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, groups=1,
                     norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, dilation=1, inplace=True, bias=None):
            if padding is None:
                padding = (kernel_size - 1) // 2 * dilation
            if bias is None:
                bias = norm_layer is None
            layers = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation, groups=groups,
                                bias=bias)]
            if norm_layer is not None:
                layers.append(norm_layer(out_channels))
            if activation_layer is not None:
                layers.append(activation_layer(inplace=inplace))
            super().__init__(*layers)
            self.out_channels = out_channels

    class SqueezeExcitation(nn.Module):
        def __init__(self, input_channels, squeeze_channels, activation=nn.ReLU, scale_activation=nn.Sigmoid):
            super().__init__()
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.fc1 = nn.Conv2d(input_channels, squeeze_channels, 1)
            self.fc2 = nn.Conv2d(squeeze_channels, input_channels, 1)
            self.activation = activation()
            self.scale_activation = scale_activation()

        def forward(self, x):
            s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
            return s * x

    def _make_divisible(v, divisor, min_value=None):
        if min_value is None:
            min_value = divisor
        new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
        if new_v < 0.9 * v:
            new_v += divisor
        return new_v

    tv = types.ModuleType("torchvision")
    tv.__path__ = []
    iru = types.ModuleType("torchvision._internally_replaced_utils")
    iru.load_state_dict_from_url = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("offline"))
    ops_pkg = types.ModuleType("torchvision.ops")
    ops_pkg.__path__ = []
    misc = types.ModuleType("torchvision.ops.misc")
    misc.ConvNormActivation, misc.SqueezeExcitation = ConvNormActivation, SqueezeExcitation
    models_pkg = types.ModuleType("torchvision.models")
    models_pkg.__path__ = []
    mu = types.ModuleType("torchvision.models._utils")
    mu._make_divisible = _make_divisible
    for name, mod in (("torchvision", tv), ("torchvision._internally_replaced_utils", iru), ("torchvision.ops", ops_pkg),
                      ("torchvision.ops.misc", misc), ("torchvision.models", models_pkg),
                      ("torchvision.models._utils", mu)):
        sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location("models.laud_regnet", os.path.join(REF_ROOT, "models", "laud_regnet.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["models.laud_regnet"] = mod
    spec.loader.exec_module(mod)
    return mod


REGNET_TINY = dict(depths=[1, 1, 2, 1], widths=[16, 32, 48, 64], group_widths=[8, 8, 8, 8],
                   bottleneck_multipliers=[1.0, 1.0, 1.0, 1.0], strides=[2, 2, 2, 2], se_ratio=0.25)
REGNET_CASES = {
    "layerskip": dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[16, 8, 4, 2]),
    "spatial_g2": dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[2, 2, 2, 1]),
    "channel_g2": dict(dyn_mode=["channel"] * 4, channel_dyn_granularity=[2, 2, 2, 2], channel_masker=["MLP"] * 4,
                       channel_masker_layers=[2, 2, 2, 2]),
    "both": dict(dyn_mode=["both"] * 4, mask_spatial_granularity=[4, 2, 2, 1], channel_dyn_granularity=[2, 2, 2, 2],
                 channel_masker=["MLP"] * 4, channel_masker_layers=[1, 1, 1, 1]),
}


def make_regnet(G):
    out = {"params": {}, "tiny_params": dict(REGNET_TINY), "cases": {}}
    for name, kw in (("lad_regnet_y_400mf", dict(depth=16, w_0=48, w_a=27.89, w_m=2.09, group_width=8)),
                     ("lad_regnet_y_800mf", dict(depth=14, w_0=56, w_a=38.84, w_m=2.4, group_width=16)),
                     ("lad_regnet_y_1_6gf", dict(depth=27, w_0=48, w_a=20.71, w_m=2.65, group_width=24)),
                     ("lad_regnet_y_3_2gf", dict(depth=21, w_0=80, w_a=42.63, w_m=2.66, group_width=24)),
                     ("lad_regnet_y_8gf", dict(depth=17, w_0=192, w_a=76.82, w_m=2.19, group_width=56)),
                     ("lad_regnet_y_16gf", dict(depth=18, w_0=200, w_a=106.23, w_m=2.48, group_width=112))):
        bp = G.BlockParams.from_init_params(se_ratio=0.25, **kw)
        out["params"][name] = dict(depths=bp.depths, widths=bp.widths, group_widths=bp.group_widths,
                                   bottleneck_multipliers=bp.bottleneck_multipliers, strides=bp.strides)
    seed, batch, size = 9, 2, 64
    for cname, kw in REGNET_CASES.items():
        bp = G.BlockParams(**REGNET_TINY)
        model = quiet(G.LAD_RegNet, bp, num_classes=10, stem_width=8, input_size=size, **kw).eval()
        model.load_state_dict(fill_state_dict(model.state_dict(), seed))
        x = seeded_randn((batch, 3, size, size), seed + 100)
        fx = dict(kw=dict(kw, num_classes=10, stem_width=8, input_size=size), seed=seed, x_seed=seed + 100, batch=batch,
                  keys=list(model.state_dict().keys()), n_params=sum(p.numel() for p in model.parameters()))
        with torch.no_grad():
            fx["masker_run"] = to_cpu(model(x, 1.0))
            blocks = [(n, m.f) for n, m in model.named_modules() if isinstance(m, G.ResBottleneckBlock)]
            masks = injected_masks_for(blocks, batch, seed=2000 + seed)
            for bname, f in blocks:
                if "spatial" in masks[bname]:
                    if cname == "layerskip":   # one bit per image
                        masks[bname]["spatial"] = masks[bname]["spatial"][:, :, :1, :1].contiguous()
                    inject(f.masker_spatial, masks[bname]["spatial"])
                if "channel" in masks[bname]:
                    inject(f.masker_channel, masks[bname]["channel"])
            fx["mask_seed"] = 2000 + seed
            fx["injected_run"] = to_cpu(model(x, 1.0))
        out["cases"][cname] = fx
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    U, R = load_reference()
    torch.save(make_l1(U), os.path.join(HERE, "l1_ops.pt"))
    seed = 200
    for stride in (1, 2):
        blocks = {}
        for name in BLOCK_CASES:
            blocks[f"{name}_s{stride}"] = make_block(U, R, name, stride, seed)
            seed += 10
        torch.save(blocks, os.path.join(HERE, f"blocks_s{stride}.pt"))
    full = {name: make_full(R, name) for name in FULL_CASES}
    torch.save(full, os.path.join(HERE, "full_tiny.pt"))
    torch.save(make_regnet(load_reference_regnet()), os.path.join(HERE, "regnet_tiny.pt"))
    for f in ("l1_ops.pt", "blocks_s1.pt", "blocks_s2.pt", "full_tiny.pt", "regnet_tiny.pt"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
