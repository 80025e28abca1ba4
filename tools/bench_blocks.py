#!/usr/bin/env python3
"""Tuning: per-block time of the headline model (R101 channel-2222, bs256, keep 0.62) under channel_exec = gather / dense,
one representative non-downsample block per stage, HIP events."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import laudnet_amd
from laudnet_amd import ops
from fill import fill_state_dict
import bench
ops.set_math_mode("bf16x3")
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["channel"]
m = laudnet_amd.uni_resnet101(**dict(wl["kw"], num_classes=1000, input_size=224)).eval()
m.load_state_dict(fill_state_dict(m.state_dict(), 1)); m = m.to(dev)
B = 256
g = torch.Generator().manual_seed(0)
shapes = {1: (256, 56), 2: (512, 28), 3: (1024, 14), 4: (2048, 7)}
with torch.no_grad():
    for st in (1, 2, 3, 4):
        C, H = shapes[st]
        blk = getattr(m, f"layer{st}")[1]
        x = torch.relu(torch.randn(B, C, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
        G = blk.masker_channel.channel_dyn_group
        blk.forced_channel_mask = (torch.rand(B, G, generator=g) < 0.62).float().to(dev)
        blk.inplace_residual = False
        for mode in ("auto", "gather", "dense"):
            blk.channel_exec = mode
            for _ in range(3): blk.run_dynamic(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): blk.run_dynamic(x)
            e1.record(); torch.cuda.synchronize()
            print(f"stage {st} {mode:7s}: {100 * e0.elapsed_time(e1):8.1f} us per block", flush=True)
