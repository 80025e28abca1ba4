#!/usr/bin/env python3
"""Tuning: distribution of active channels per image (K_b) of every block of the bench model after calibration."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import laudnet_amd
from laudnet_amd import ops
from fill import fill_state_dict, seeded_randn
import bench
ops.set_math_mode("bf16x3")
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["channel"]
m = laudnet_amd.uni_resnet101(**dict(wl["kw"], num_classes=1000, input_size=224)).eval()
sd = fill_state_dict(m.state_dict(), 1)
for k in sd:
    if k.endswith("bn3.weight"): sd[k] = sd[k] * 0.3
m.load_state_dict(sd); m = m.to(dev)
x = seeded_randn((256, 3, 224, 224), 1000).to(dev).contiguous(memory_format=torch.channels_last)
bench.calibrate_maskers(m, x, 0.62, None)
with torch.no_grad():
    m(x, 1.0)
for s in (1, 2, 3, 4):
    for i, blk in enumerate(getattr(m, f"layer{s}")):
        c = blk.last_channel_cnt.float().cpu()
        W = blk.width
        print(f"stage {s} block {i:2d}: W {W}  mean {c.mean():6.1f}  min {c.min():5.0f}  max {c.max():5.0f}  >0.75W: {(c > 0.75 * W).sum().item():3d}  chunks(mean/max) {torch.ceil(c / 32).mean():.2f}/{torch.ceil(c / 32).max():.0f}")
