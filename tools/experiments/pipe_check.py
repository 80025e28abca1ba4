#!/usr/bin/env python3
"""Batch pipelining (LDN_BATCH_PIPE): the pipelined forward equals the single-stream forward bit for bit (per-image kernels)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import laudnet_amd
from laudnet_amd import ops
import bench
from fill import fill_state_dict, seeded_randn
dev = torch.device("cuda:0")
ops.set_math_mode("bf16x3")
for wname in ("channel", "spatial", "layer"):
    wl = bench.WORKLOADS[wname]
    model = laudnet_amd.uni_resnet101(**dict(wl["kw"], num_classes=1000, input_size=224)).eval()
    sd = fill_state_dict(model.state_dict(), 1)
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.3
    model.load_state_dict(sd); model = model.to(dev)
    x = seeded_randn((64, 3, 224, 224), 1000).to(dev).contiguous(memory_format=torch.channels_last)
    bench.calibrate_maskers(model, x, wl["p_channel"], wl["p_spatial"])
    outs = {}
    for n in (1, 2, 4):
        model.batch_pipeline = n
        with torch.no_grad():
            outs[n] = model(x, 1.0)
        torch.cuda.synchronize()
    for n in (2, 4):
        eq = torch.equal(outs[1][0], outs[n][0])
        st = max(float((torch.cat(a) - torch.cat(b)).abs().max()) for a, b in zip(outs[1][1:5], outs[n][1:5]))
        print(wname, "parts", n, "logits bit-identical:", eq, "max |logit diff|", float((outs[1][0] - outs[n][0]).abs().max()),
              "max stat diff", st, "flops rel", float(abs(outs[1][6] - outs[n][6]) / outs[1][6]))
