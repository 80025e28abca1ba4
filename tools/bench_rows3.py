#!/usr/bin/env python3
"""Tuning: the three launches of a spatial-mode bottleneck at the R101 stage shapes (bs256, patch masks of the canonical S=4-4-2-1
granularity, keep p), round 4's un-split kernels against round 5's pre-split path (k_dense<OF> -> k_rows3 -> k_dense<PS>).
usage: python tools/bench_rows3.py [--stage 3,2,1] [--p 0.5] [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from laudnet_amd import ops  # noqa: E402
from fill import seeded_bernoulli  # noqa: E402

STAGES = {1: (56, 256, 64, 14), 2: (28, 512, 128, 7), 3: (14, 1024, 256, 7), 4: (7, 2048, 512, 7)}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="3,2,1,4")
    ap.add_argument("--p", type=float, default=0.5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_math_mode("bf16x3")
    B = args.batch
    for st in [int(s) for s in args.stage.split(",")]:
        H, Cin, W, S = STAGES[st]
        patch = seeded_bernoulli((B, S, S), args.p, 5).to(dev)
        ix = ops.mask_to_index(patch, H, H, 1)
        n3, n1 = int(ix.cnt[0]), int(ix.cnt[1])
        x = torch.relu(torch.randn(B * H * H, Cin, device=dev))
        w1 = torch.randn(W, 1, Cin, device=dev) * 0.05
        w2 = torch.randn(W, 9, W, device=dev) * 0.05
        w3 = torch.randn(Cin, 1, W, device=dev) * 0.05
        sW, tW = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1
        tC = torch.randn(Cin, device=dev) * 0.1
        h1 = torch.empty(ix.cap1, W, device=dev)
        h2 = torch.empty(ix.cap3, W, device=dev)
        out = x.clone()
        old = {
            "conv1": lambda: ops.conv_rows(x, w1, sW, tW, h1, a_rows=ix.idx1, taps=1, m_count=ix.cnt[1:2], m_cap=ix.cap1, rows_hint=n1),
            "conv2": lambda: ops.conv_rows(h1, w2, sW, tW, h2, a_rows=ix.nbr, taps=9, m_count=ix.cnt[0:1], m_cap=ix.cap3, rows_hint=n3),
            "conv3": lambda: ops.conv_rows(h2, w3, None, tC, out, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_rows=ix.idx3, residual2d=out, rows_hint=n3),
        }
        new = {
            "conv1": lambda: ops.conv_rows_ps(x, w1, sW, tW, h1, out_presplit=True, a_rows=ix.idx1, m_count=ix.cnt[1:2], m_cap=ix.cap1, rows_hint=n1),
            "conv2": lambda: ops.conv3x3_rows_ps(h1, ix.nbr, w2, sW, tW, h2, m_count=ix.cnt[0:1], m_cap=ix.cap3, out_presplit=True, rows_hint=n3),
            "conv3": lambda: ops.conv_rows_ps(h2, w3, None, tC, out, a_presplit=True, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_rows=ix.idx3, residual2d=out, rows_hint=n3),
        }
        print(f"stage {st}: H {H} Cin {Cin} W {W}  rows3 {n3} rows1 {n1}", flush=True)
        tot_o = tot_n = 0.0
        for k in ("conv1", "conv2", "conv3"):
            to, tn = timeit(old[k], args.iters), timeit(new[k], args.iters)
            tot_o += to
            tot_n += tn
            extra = ""
            if k == "conv2":
                fl = 2.0 * n3 * 9 * W * W
                extra = f"   new: {3 * fl / tn / 1e6:7.1f} TF/s executed = {3 * fl / tn / 1e6 / 2500:.3f} of the bf16 peak"
            print(f"   {k}: round 4 {to:7.1f} us   pre-split {tn:7.1f} us{extra}", flush=True)
        print(f"   sum:   round 4 {tot_o:7.1f} us   pre-split {tot_n:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
