#!/usr/bin/env python3
"""Pins the mmdetection-3.3.0 twin of the reference's detection backbone to the 2.21.0 one.

RUNS /root/reference/mmdetection-3.3.0/mmdet/models/backbones/lad_mmdet_resnet.py (imported by path, unmodified) on the SAME seeded
cases, state dicts, inputs and injected masks as make_det_golden.py (which runs the 2.21.0 file) and writes det_tiny_330.json:
per case `equal_to_2_21_0` -- the largest absolute difference of every output (stage maps, sparsity lists, flops, dense_flops)
from the committed det_tiny.pt -- and float64 checksums of the twin's outputs (numbers only).  Build container only (needs /root/reference).  mmcv / mmengine / mmdet.registry / mmdet.utils are not
installed: builder-written stand-ins with the constructors' semantics (as in make_det_golden.py), so the twin is pinned MODULO them.
The two files differ (diff of the two sources) only in imports, the registry decorator, `forward(x)` without the unused
(iter_now, len_loader), the plugin loop variable, and the selection of `out_indices` at the end (all four stages by default)."""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_det_golden as D2  # noqa: E402  (cases, seeds, helpers; its loader is NOT called here)
from fill import fill_state_dict  # noqa: E402

REF = "/root/reference/mmdetection-3.3.0/mmdet/models"


def load_reference_330():
    mmcv, cnn = types.ModuleType("mmcv"), types.ModuleType("mmcv.cnn")
    mmengine, model = types.ModuleType("mmengine"), types.ModuleType("mmengine.model")

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
    mmcv.cnn = cnn
    model.BaseModule, model.Sequential = BaseModule, Sequential
    mmengine.model = model
    mmdet, registry, mutils = types.ModuleType("mmdet"), types.ModuleType("mmdet.registry"), types.ModuleType("mmdet.utils")

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls
    registry.MODELS = _Registry()
    mutils.ConfigType = mutils.OptConfigType = mutils.OptMultiConfig = object
    mmdet.registry, mmdet.utils = registry, mutils
    mmdet.__path__ = []
    sys.modules.update({"mmcv": mmcv, "mmcv.cnn": cnn, "mmengine": mmengine, "mmengine.model": model, "mmdet": mmdet,
                        "mmdet.registry": registry, "mmdet.utils": mutils})

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

    pkg("refdet330")
    layers = pkg("refdet330.layers")
    layers.LAD_MMDet_Reslayer = load("refdet330.layers.res_layer", os.path.join(REF, "layers", "res_layer.py")).LAD_MMDet_Reslayer
    pkg("refdet330.backbones", os.path.join(REF, "backbones"))
    load("refdet330.backbones.utils", os.path.join(REF, "backbones", "utils.py"))
    return load("refdet330.backbones.lad_mmdet_resnet", os.path.join(REF, "backbones", "lad_mmdet_resnet.py"))


def max_diff(a, b):
    if torch.is_tensor(a):
        return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0
    if isinstance(a, dict):
        assert sorted(a) == sorted(b)
        return max([max_diff(a[k], b[k]) for k in a] + [0.0])
    if isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        return max([max_diff(x, y) for x, y in zip(a, b)] + [0.0])
    return 0.0 if a == b else float("inf")


def checksums(o, acc=None):
    acc = [] if acc is None else acc
    if torch.is_tensor(o):
        acc.append([float(o.double().sum()), float(o.double().abs().sum())])
    elif isinstance(o, dict):
        for k in sorted(o):
            checksums(o[k], acc)
    elif isinstance(o, (list, tuple)):
        for v in o:
            checksums(v, acc)
    elif isinstance(o, (int, float)):
        acc.append([float(o), abs(float(o))])
    return acc


def checksum_keys(keys):
    import hashlib
    return hashlib.sha256("\n".join(keys).encode()).hexdigest()


def main():
    R = load_reference_330()
    want = torch.load(os.path.join(HERE, "det_tiny.pt"), weights_only=False)
    out = {}
    for name, (kw, shape) in D2.CASES.items():
        seed = 11
        model = R.LAD_MMDet_ResNet(**kw)
        model.eval()
        assert list(model.state_dict().keys()) == want[name]["keys"], "the twin must expose the same state_dict keys"
        model.load_state_dict(fill_state_dict(model.state_dict(), seed))
        x = D2.seeded_randn(shape, seed + 100)
        fx = dict(kw=kw, shape=shape, seed=seed, x_seed=seed + 100, keys=list(model.state_dict().keys()),
                  n_params=sum(p.numel() for p in model.parameters()))
        with torch.no_grad():
            fx["masker_run"] = D2.to_cpu(model(x))
            i = 0
            for s in (1, 2, 3, 4):
                for j, blk in enumerate(getattr(model, f"layer{s}")):
                    m = want[name]["masks"][f"layer{s}.{j}"]
                    if "channel" in m:
                        D2.inject(blk.masker_channel, m["channel"])
                    else:
                        D2.inject(blk.masker_spatial, m["spatial"])
                    i += 1
            fx["injected_run"] = D2.to_cpu(model(x))
        fx["equal_to_2_21_0"] = dict(masker_run=max_diff(fx["masker_run"], want[name]["masker_run"]),
                                     injected_run=max_diff(fx["injected_run"], want[name]["injected_run"]))
        out[name] = fx
        print(name, "max |3.3.0 - 2.21.0|:", fx["equal_to_2_21_0"])
    # the twin's outputs are stored as checksums only (they are bit-equal to det_tiny.pt, which holds the tensors): per run the
    # flattened list of (sum, sum of absolute values) in float64 of every tensor, in traversal order
    import json
    js = {name: dict(keys_sha=checksum_keys(fx["keys"]), n_params=fx["n_params"], equal_to_2_21_0=fx["equal_to_2_21_0"],
                     masker_run=checksums(fx["masker_run"]), injected_run=checksums(fx["injected_run"])) for name, fx in out.items()}
    json.dump(js, open(os.path.join(HERE, "det_tiny_330.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
