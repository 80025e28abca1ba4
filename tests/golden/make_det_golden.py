#!/usr/bin/env python3
"""Generates tests/golden/det_tiny.pt by RUNNING the reference's detection backbone
(/root/reference/mmdetection-2.21.0/mmdet/models/backbones/lad_mmdet_resnet.py, imported by path, unmodified) on seeded inputs.
Build container only (needs /root/reference).  mmcv is not installed here: the four mmcv symbols the three reference files use
are builder-written stand-ins with mmcv-1.x semantics (build_conv_layer(None, ...) = nn.Conv2d, build_norm_layer(dict(type='BN'),
n, postfix) = ('bn<postfix>', nn.BatchNorm2d(n)), BaseModule = nn.Module keeping init_cfg, Sequential = BaseModule + nn.Sequential),
so the fixture is pinned to the reference MODULO these stubs.  Only tensors / plain containers are stored."""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from fill import fill_state_dict  # noqa: E402

REF = "/root/reference/mmdetection-2.21.0/mmdet/models"


def seeded_randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def seeded_bernoulli(shape, p, seed):
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) < p).float()


def load_reference():
    mmcv = types.ModuleType("mmcv")
    cnn = types.ModuleType("mmcv.cnn")
    runner = types.ModuleType("mmcv.runner")

    def build_conv_layer(cfg, *args, **kwargs):
        assert cfg is None
        return nn.Conv2d(*args, **kwargs)

    def build_norm_layer(cfg, num_features, postfix=""):
        assert cfg["type"] == "BN"
        layer = nn.BatchNorm2d(num_features, eps=cfg.get("eps", 1e-5))
        for p in layer.parameters():
            p.requires_grad = cfg.get("requires_grad", True)
        return "bn" + str(postfix), layer

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    class Sequential(BaseModule, nn.Sequential):
        def __init__(self, *args, init_cfg=None):
            BaseModule.__init__(self, init_cfg)
            nn.Sequential.__init__(self, *args)

    cnn.build_conv_layer, cnn.build_norm_layer, cnn.build_plugin_layer = build_conv_layer, build_norm_layer, None
    runner.BaseModule, runner.Sequential = BaseModule, Sequential
    mmcv.cnn, mmcv.runner = cnn, runner
    sys.modules.update({"mmcv": mmcv, "mmcv.cnn": cnn, "mmcv.runner": runner})

    def pkg(name, path=None):
        m = types.ModuleType(name)
        m.__path__ = [path] if path else []
        sys.modules[name] = m
        return m

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    pkg("refdet")
    builder = pkg("refdet.builder")

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls
    builder.BACKBONES = _Registry()
    utils = pkg("refdet.utils")
    utils.LAD_MMDet_Reslayer = load("refdet.utils.res_layer", os.path.join(REF, "utils", "res_layer.py")).LAD_MMDet_Reslayer
    pkg("refdet.backbones", os.path.join(REF, "backbones"))
    load("refdet.backbones.utils", os.path.join(REF, "backbones", "utils.py"))
    return load("refdet.backbones.lad_mmdet_resnet", os.path.join(REF, "backbones", "lad_mmdet_resnet.py"))


CASES = {
    # name: (constructor kwargs, input [B, 3, H, W])
    "channel_r50": (dict(depth=50, base_channels=16, dyn_mode=["channel"] * 4, channel_dyn_granularity=[2, 2, 2, 2],
                         channel_masker=["MLP"] * 4, channel_masker_layers=[2, 2, 2, 2], temperature_0=1.0, sparsity_target=0.5),
                    (2, 3, 64, 96)),
    "layer_r50": (dict(depth=50, base_channels=16, dyn_mode=["layer"] * 4, temperature_0=1.0, sparsity_target=0.5), (3, 3, 96, 64)),
    "channel_conv_linear_r50": (dict(depth=50, base_channels=16, dyn_mode=["channel"] * 4, channel_dyn_granularity=[4, 4, 4, 4],
                                     channel_masker=["conv_linear"] * 4, temperature_0=1.0), (2, 3, 32, 64)),
}


def to_cpu(o):
    if torch.is_tensor(o):
        return o.detach().cpu().clone()
    if isinstance(o, dict):
        return {k: to_cpu(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [to_cpu(v) for v in o]
    return o


def inject(masker, mask):
    flops_of = masker.forward

    def fwd(x, temperature):
        _, _, fl = flops_of(x, temperature)
        return mask, mask.mean(), fl
    masker.forward = fwd


def main():
    R = load_reference()
    out = {}
    for name, (kw, shape) in CASES.items():
        seed = 11
        model = R.LAD_MMDet_ResNet(**kw)
        model.eval()            # (the reference's train() override returns None)
        model.load_state_dict(fill_state_dict(model.state_dict(), seed))
        x = seeded_randn(shape, seed + 100)
        fx = dict(kw=kw, shape=shape, seed=seed, x_seed=seed + 100, keys=list(model.state_dict().keys()),
                  n_params=sum(p.numel() for p in model.parameters()))
        with torch.no_grad():
            fx["masker_run"] = to_cpu(model(x))
            masks = {}
            i = 0
            for s in (1, 2, 3, 4):
                for j, blk in enumerate(getattr(model, f"layer{s}")):
                    bname = f"layer{s}.{j}"
                    if hasattr(blk, "masker_channel"):
                        m = seeded_bernoulli((shape[0], blk.masker_channel.channel_dyn_group), 0.62, 2000 + 2 * i + 1)
                        inject(blk.masker_channel, m)
                        masks[bname] = {"channel": m}
                    else:
                        m = seeded_bernoulli((shape[0], 1, 1, 1), 0.5, 2000 + 2 * i)
                        inject(blk.masker_spatial, m)
                        masks[bname] = {"spatial": m}
                    i += 1
            fx["masks"] = masks
            fx["injected_run"] = to_cpu(model(x))
        out[name] = fx
        print(name, [tuple(t.shape) for t in fx["masker_run"][0]], "flops", float(fx["masker_run"][1]["flops"]),
              "dense", float(fx["masker_run"][1]["dense_flops"]))
    torch.save(out, os.path.join(HERE, "det_tiny.pt"))


if __name__ == "__main__":
    main()
