#!/usr/bin/env python3
"""Damped twins of full_tiny.pt: the SAME eight tiny full models run by the imported reference (make_golden.load_reference), with
the residual branches damped (every bn3.weight * 0.3 -- the recipe of bench.py and tests/test_hip_fullsize.py) so that 50 / 101
seeded-random layers keep O(1) logits.  The consumers can then assert the north star's PLAIN 1e-3 on the logits, with no slack
that scales with the logits (test hygiene (a)).  Build container only (needs /root/reference):

    python tests/golden/make_full_damped_golden.py

Tensors and plain-python config only; parameters come from fill.fill_state_dict + fill.damp_residual_branches by seed."""
from __future__ import annotations

import os

import torch

import make_golden as MG
from fill import damp_residual_branches, fill_state_dict, seeded_randn

HERE = os.path.dirname(os.path.abspath(__file__))
DAMP = 0.3


def make_full_damped(R, name, seed=11, batch=2, width_mult=0.125, input_size=224):
    factory, kw = MG.FULL_CASES[name]
    kw = dict(kw, width_mult=width_mult, input_size=input_size, num_classes=1000)
    model = MG.quiet(getattr(R, factory), **kw).eval()
    model.load_state_dict(damp_residual_branches(fill_state_dict(model.state_dict(), seed), DAMP))
    x = seeded_randn((batch, 3, input_size, input_size), seed + 100)
    fx = dict(factory=factory, kw=kw, seed=seed, x_seed=seed + 100, batch=batch, damp=DAMP, keys=list(model.state_dict().keys()))
    with torch.no_grad():
        fx["masker_run"] = MG.to_cpu(model(x, 1.0))
        blocks = [(f"layer{s}.{j}", b) for s in (1, 2, 3, 4) for j, b in enumerate(getattr(model, f"layer{s}"))]
        masks = MG.injected_masks_for(blocks, batch, seed=1000 + seed)
        for bname, blk in blocks:
            if "spatial" in masks[bname]:
                MG.inject(blk.masker_spatial, masks[bname]["spatial"])
            if "channel" in masks[bname]:
                MG.inject(blk.masker_channel, masks[bname]["channel"])
        fx["mask_seed"] = 1000 + seed
        fx["injected_run"] = MG.to_cpu(model(x, 1.0))
    return fx


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    _, R = MG.load_reference()
    full = {name: make_full_damped(R, name) for name in MG.FULL_CASES}
    for name, fx in full.items():
        print(name, "max |logit| masker %.3f injected %.3f" % (fx["masker_run"][0].abs().max(), fx["injected_run"][0].abs().max()))
    torch.save(full, os.path.join(HERE, "full_tiny_damped.pt"))
    print("full_tiny_damped.pt", os.path.getsize(os.path.join(HERE, "full_tiny_damped.pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
