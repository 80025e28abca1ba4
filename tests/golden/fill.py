"""Deterministic, construction-order-independent parameter fill.

Used by BOTH tests/golden/make_golden.py (on the reference's modules, in the build
container) and the tests (on the oracle / the HIP-backed modules): every tensor of
a state_dict is drawn from a torch CPU Generator seeded by crc32(key) ^ seed, so a
model never has to be shipped inside a fixture -- only `seed` and the expected
outputs are.  Distributions follow SURVEY 8c/8d: He-normal convs, non-trivial BN
eval statistics (so the pre-BN channel-mask constants of laud_resnet.py:116-117
are non-zero), N(0,1) masker weights with zero bias (~50 % dense masks).
"""
from __future__ import annotations

import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_state_dict(template: dict, seed: int) -> dict:
    """Return a new state_dict with the same keys/shapes/dtypes as `template`."""
    out = {}
    for key, ref in template.items():
        g = _gen(key, seed)
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros(shape, dtype=ref.dtype)
            continue
        in_masker = "masker" in key
        is_bn = leaf in ("running_mean", "running_var") or (ref.dim() == 1 and _looks_like_bn(key, template))
        if is_bn:
            if leaf == "running_mean":
                t = torch.randn(shape, generator=g) * 0.1
            elif leaf == "running_var":
                t = torch.rand(shape, generator=g) + 0.5
            elif leaf == "weight":
                t = torch.rand(shape, generator=g) + 0.5
            else:  # bias
                t = torch.randn(shape, generator=g) * 0.1
        elif in_masker:
            t = torch.randn(shape, generator=g) if leaf == "weight" else torch.zeros(shape)
        elif ref.dim() == 4:  # conv: He-normal fan_out
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * (2.0 / fan_out) ** 0.5
        elif ref.dim() == 2:  # classifier
            t = torch.randn(shape, generator=g) * 0.01
        else:  # plain bias
            t = torch.randn(shape, generator=g) * 0.01
        out[key] = t.to(ref.dtype)
    return out


def _looks_like_bn(key: str, template: dict) -> bool:
    prefix = key.rsplit(".", 1)[0]
    return (prefix + ".running_mean") in template


def seeded_randn(shape, seed: int) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(seed))


def seeded_bernoulli(shape, p: float, seed: int) -> torch.Tensor:
    u = torch.rand(tuple(shape), generator=torch.Generator().manual_seed(seed))
    return (u < p).to(torch.float32)


def damp_residual_branches(sd: dict, factor: float = 0.3) -> dict:
    """Scale the last BN of every residual branch (ResNet bn3 / LAD-RegNet conv c's BN) so that tens of seeded-random blocks keep
    O(1) activations and logits (what zero_init_residual is for); the recipe of bench.py and tests/test_hip_fullsize.py."""
    return {k: (v * factor if k.endswith("bn3.weight") or k.endswith(".f.c.1.weight") else v) for k, v in sd.items()}
