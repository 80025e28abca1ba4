#!/usr/bin/env python3
"""One training step (forward + backward, frozen BatchNorm statistics, Gumbel-hard masks, sparsity criterion) of a full-width LAUD-ResNet on the row
kernels (laudnet_amd.training.train_forward) beside the oracle's dense emulation run through PyTorch on the same GPU.  One JSON line per workload.
usage: tools/bench_train.py [--batch 32] [--steps 5] [--workloads layer,spatial,channel]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import bench  # noqa: E402
import laudnet_amd  # noqa: E402
from fill import fill_state_dict, seeded_randn  # noqa: E402
from laudnet_amd import ops  # noqa: E402
from laudnet_amd.training import prepare_for_training, train_forward  # noqa: E402
from oracle import torch_ref as TR  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--workloads", default="layer,spatial,channel")
ap.add_argument("--math", default="bf16x3")
args = ap.parse_args()
dev = torch.device("cuda", 0)
ops.set_math_mode(args.math)
for w in args.workloads.split(","):
    wl = bench.WORKLOADS[w]
    kw = dict(wl["kw"], num_classes=1000, input_size=224)
    hip = laudnet_amd.uni_resnet101(**kw)
    sd = fill_state_dict(hip.state_dict(), 1)
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.3
    hip.load_state_dict(sd)
    hip = hip.to(dev).eval()
    x = seeded_randn((args.batch, 3, 224, 224), 1000).to(dev).contiguous(memory_format=torch.channels_last)
    bench.calibrate_maskers(hip, x, wl["p_channel"], wl["p_spatial"])
    sd = {k: v.detach().clone() for k, v in hip.state_dict().items()}
    ref = TR.resnet101_ref(**kw)
    ref.load_state_dict(sd)
    ref = ref.to(dev).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    prepare_for_training(hip)
    g = seeded_randn((args.batch, 1000), 5).to(dev)

    def step(fwd, model):
        for p in model.parameters():
            p.grad = None
        out = fwd()
        loss = (out[0] * g).sum() / 10.0 + 10.0 * (out[5].mean() - 0.5) ** 2
        loss.backward()
        return out

    res = {}
    for name, fwd, model in (("hip_row_kernels", lambda: train_forward(hip, x, 1.0), hip), ("dense_emulation_pytorch", lambda: ref(x, 1.0), ref)):
        torch.manual_seed(3)
        for _ in range(2):
            out = step(fwd, model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(fwd, model)
        torch.cuda.synchronize()
        res[name] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / args.steps, "mean_block_flops_ratio": round(float(out[5].detach().mean()), 4)}
    res["speedup"] = res["dense_emulation_pytorch"]["ms_per_step"] / res["hip_row_kernels"]["ms_per_step"]
    print(json.dumps({"workload": wl["name"], "batch": args.batch, "steps": args.steps, "math": args.math,
                      "what": "one training step = forward + backward of every parameter, frozen BatchNorm statistics, Gumbel-hard masks", **res}), flush=True)
    del hip, ref
    torch.cuda.empty_cache()
