#!/usr/bin/env python3
"""Kernel tuning harness: times ldn_conv_image on the R101 channel-2222 layer shapes (bs256, keep-prob 0.62)
with HIP events and prints achieved TFLOP/s and GB/s per layer kind.

usage: python tools/bench_conv.py [--stage 1,2,3,4] [--kinds conv1,conv2,conv3] [--iters 10] [--batch 256]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import ops  # noqa: E402

STAGES = {1: (56, 256, 64), 2: (28, 512, 128), 3: (14, 1024, 256), 4: (7, 2048, 512)}  # H, Cin(rest), W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="1,2,3,4")
    ap.add_argument("--kinds", default="conv1,conv2,conv3")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--p", type=float, default=0.62)
    ap.add_argument("--gran", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    g = torch.Generator().manual_seed(0)
    for st in [int(s) for s in args.stage.split(",")]:
        H, Cin, W = STAGES[st]
        G = W // args.gran
        gm = (torch.rand(B, G, generator=g) < args.p).float().to(dev)
        _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, args.gran, mask_in=gm)
        cntf = cnt.double().cpu()
        x = torch.randn(B, H, H, Cin, device=dev)
        h1 = torch.randn(B, H, H, W, device=dev)
        h2 = torch.randn(B, H, H, W, device=dev)
        out = torch.empty(B, H, H, Cin, device=dev)
        w1 = torch.randn(W, 1, Cin, device=dev) * 0.05
        w2 = torch.randn(9, W, W, device=dev) * 0.05      # k-major [taps][cin][cout]
        w3 = torch.randn(1, W, Cin, device=dev) * 0.05    # k-major
        sW, tW = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1
        sC, tC = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
        tab = torch.randn(16, W, device=dev) * 0.1
        cW = torch.rand(W, device=dev) * 0.1
        hw = H * H
        runs = {
            "conv1": (lambda: ops.conv_image(x, w1, sW, tW, h1, n_idx=idx, n_cnt=cnt, post_sub=cW, relu=1),
                      float((2.0 * hw * Cin * cntf).sum()), 4.0 * B * hw * Cin + 4.0 * hw * float(cntf.sum())),
            "conv2": (lambda: ops.conv_image(h1, w2, sW, tab, h2, ksize=3, stride=1, k_idx=idx, k_cnt=cnt, kgran=args.gran,
                                             n_idx=idx, n_cnt=cnt, post_sub=cW, relu=1),
                      float((2.0 * hw * 9 * cntf * cntf).sum()), 8.0 * hw * float(cntf.sum())),
            "conv3": (lambda: ops.conv_image(h2, w3, None, tC, out, k_idx=idx, k_cnt=cnt, kgran=args.gran, relu=1, residual=x),
                      float((2.0 * hw * Cin * cntf).sum()), 8.0 * B * hw * Cin + 4.0 * hw * float(cntf.sum())),
        }
        # dense counterparts (shared n-major weights, no channel lists): what the same layers cost without skipping
        wd1 = torch.randn(W, 1, Cin, device=dev) * 0.05
        wd2 = torch.randn(W, 9, W, device=dev) * 0.05
        wd3 = torch.randn(Cin, 1, W, device=dev) * 0.05
        runs.update({
            "dense1": (lambda: ops.conv_image(x, wd1, sW, tW, h1, relu=1), 2.0 * B * hw * Cin * W, 4.0 * B * hw * (Cin + W)),
            "dense2": (lambda: ops.conv_image(h1, wd2, sW, tab, h2, ksize=3, stride=1, relu=1), 2.0 * B * hw * 9 * W * W, 8.0 * B * hw * W),
            "dense3": (lambda: ops.conv_image(h2, wd3, sC, tC, out, relu=1, residual=x), 2.0 * B * hw * Cin * W, 4.0 * B * hw * (2 * Cin + W)),
        })
        for kind in args.kinds.split(","):
            fn, flops, bytes_ = runs[kind]
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            print(f"stage{st} {kind}: {us:8.1f} us  {flops / us / 1e6:7.2f} TFLOP/s  {bytes_ / us / 1e3:8.1f} GB/s (min-traffic)  "
                  f"[{flops / 1e9:.1f} GFLOP, {bytes_ / 1e6:.0f} MB]", flush=True)


if __name__ == "__main__":
    main()
