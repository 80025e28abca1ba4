#!/usr/bin/env python3
"""Tuning: shared-weight packed-row convs (ldn_conv_rows, all pixels active) on the R101 stage shapes -- what a
channel-masked stage costs when it is executed densely over multi-image M tiles instead of per-image gathered GEMMs.
usage: python tools/bench_rows.py [--stage 4,3] [--iters 10]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import ops  # noqa: E402

STAGES = {1: (56, 256, 64), 2: (28, 512, 128), 3: (14, 1024, 256), 4: (7, 2048, 512)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="4,3,2,1")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    for st in [int(s) for s in args.stage.split(",")]:
        H, Cin, W = STAGES[st]
        rows = B * H * H
        ix = ops.mask_to_index(torch.ones(B, H, H, device=dev), H, H, 1)
        x = torch.randn(rows, Cin, device=dev)
        h1 = torch.empty(rows, W, device=dev)
        h2 = torch.empty(rows, W, device=dev)
        out = torch.empty(rows, Cin, device=dev)
        w1 = torch.randn(W, 1, Cin, device=dev) * 0.05
        w2 = torch.randn(W, 9, W, device=dev) * 0.05
        w3 = torch.randn(Cin, 1, W, device=dev) * 0.05
        sW, tW = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1
        sC, tC = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
        runs = {
            "rows1": (lambda: ops.conv_rows(x, w1, sW, tW, h1, taps=1, m_cap=rows), 2.0 * rows * Cin * W),
            "rows2": (lambda: ops.conv_rows(h1, w2, sW, tW, h2, a_rows=ix.nbr, taps=9, m_cap=rows), 2.0 * rows * 9 * W * W),
            "rows3": (lambda: ops.conv_rows(h2, w3, sC, tC, out, taps=1, m_cap=rows, residual2d=x), 2.0 * rows * Cin * W),
        }
        for kind, (fn, flops) in runs.items():
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
            print(f"stage{st} {kind}: {us:8.1f} us  {flops / us / 1e6:7.2f} TFLOP/s  [{flops / 1e9:.1f} GFLOP]", flush=True)


if __name__ == "__main__":
    main()
