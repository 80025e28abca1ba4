"""Shared test helpers (CPU and GPU tests)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from fill import fill_state_dict, seeded_bernoulli, seeded_randn  # tests/golden/fill.py

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def block_input(fx):
    return F.relu(seeded_randn(fx["x_shape"], fx["x_seed"]))


def start_state(x):
    return (x, None, None, None, None, None, torch.tensor(0.0, device=x.device))


def make_block(block_cls, fx, conv_cls=nn.Conv2d):
    """Instantiate `block_cls` (oracle BottleneckRef or the HIP-backed Bottleneck) from a block fixture."""
    kw = dict(fx["kw"])
    down = None
    if fx["has_downsample"]:
        down = nn.Sequential(nn.Conv2d(kw["inplanes"], kw["planes"] * 4, 1, stride=kw["stride"], bias=False),
                             nn.BatchNorm2d(kw["planes"] * 4))
    blk = block_cls(downsample=down, **kw).eval()
    blk.load_state_dict(fill_state_dict(blk.state_dict(), fx["seed"]))
    return blk


def full_model_blocks(model):
    return [(f"layer{s}.{j}", b) for s in (1, 2, 3, 4) for j, b in enumerate(getattr(model, f"layer{s}"))]


def injected_masks_for(blocks, batch, seed, p_spatial=0.5, p_channel=0.62):
    """Same recipe as tests/golden/make_golden.py:injected_masks_for (kept in step by
    test_oracle_golden.py::test_full_models_injected)."""
    masks = {}
    for i, (name, blk) in enumerate(blocks):
        entry = {}
        if blk.masker_spatial is not None:
            entry["spatial"] = seeded_bernoulli((batch, blk.masker_spatial.groups, blk.masker_spatial.mask_size,
                                                 blk.masker_spatial.mask_size), p_spatial, seed + 2 * i)
        if blk.masker_channel is not None:
            entry["channel"] = seeded_bernoulli((batch, blk.masker_channel.groups), p_channel, seed + 2 * i + 1)
        masks[name] = entry
    return masks


def assert_tuple_close(got, want, atol, rtol=0.0, what=""):
    assert len(got) == len(want), what
    for i, (g, w) in enumerate(zip(got, want)):
        if isinstance(w, (list, tuple)):
            assert_tuple_close(g, w, atol, rtol, f"{what}[{i}]")
        else:
            g = torch.as_tensor(g).detach().float().cpu()
            w = torch.as_tensor(w).detach().float().cpu()
            assert g.shape == w.shape, f"{what}[{i}] shape {tuple(g.shape)} vs {tuple(w.shape)}"
            err = (g - w).abs()
            bound = atol + rtol * w.abs()
            assert bool((err <= bound).all()), f"{what}[{i}] max err {err.max().item():.3e} (atol {atol}, rtol {rtol})"
