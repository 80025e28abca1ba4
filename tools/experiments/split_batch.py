#!/usr/bin/env python3
"""Experiment (round 6): does running the two halves of a batch as two independent forwards on two streams -- two replicas of the module tree, same
weights -- hide the latency-bound launches of the spatial / layer path (k_plan: 23 us x 33 per forward, the stand-alone maskers) under the other
half's row kernels?   usage: tools/experiments/split_batch.py [--workload spatial] [--parts 2] [--steps 20]"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import bench  # noqa: E402
import laudnet_amd  # noqa: E402
from fill import fill_state_dict, seeded_randn  # noqa: E402
from laudnet_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="spatial")
ap.add_argument("--parts", type=int, default=2)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
ops.set_math_mode("bf16x3")
wl = bench.WORKLOADS[args.workload]
kw = dict(wl["kw"], num_classes=1000, input_size=224)
model = getattr(laudnet_amd, wl.get("arch", "uni_resnet101"))(**kw)
sd = fill_state_dict(model.state_dict(), 1)
for k in sd:
    if k.endswith("bn3.weight") or k.endswith(".f.c.1.weight"):
        sd[k] = sd[k] * 0.3
model.load_state_dict(sd)
model = model.to(dev).eval()
x = seeded_randn((args.batch, 3, 224, 224), 1000).to(dev).contiguous(memory_format=torch.channels_last)
bench.calibrate_maskers(model, x, wl.get("p_channel"), wl.get("p_spatial"))


def timed(fn, steps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


with torch.no_grad():
    whole = model(x, 1.0)[0].clone()
    t_whole = timed(lambda: model(x, 1.0), args.steps)
    reps = [model] + [copy.deepcopy(model) for _ in range(args.parts - 1)]
    streams = [torch.cuda.Stream(dev) for _ in range(args.parts)]
    xs = [c.contiguous(memory_format=torch.channels_last) for c in x.chunk(args.parts)]
    outs = [None] * args.parts

    def split():
        cur = torch.cuda.current_stream(dev)
        for i in range(args.parts):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs[i] = reps[i](xs[i], 1.0)
        for s in streams:
            cur.wait_stream(s)

    split()
    torch.cuda.synchronize()
    got = torch.cat([o[0] for o in outs])
    t_split = timed(split, args.steps)
    t_seq = timed(lambda: [reps[i](xs[i], 1.0) for i in range(args.parts)], args.steps)
print(json.dumps({"workload": args.workload, "batch": args.batch, "parts": args.parts, "ms_whole_batch": t_whole, "ms_parts_on_streams": t_split,
                  "ms_parts_one_stream": t_seq, "max_abs_logit_diff_vs_whole": float((got - whole).abs().max())}))
