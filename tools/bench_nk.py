#!/usr/bin/env python3
"""Tuning: the n-major (no channel list) 1x1 launches of the headline forward, timed with HIP events.
LDN_STREAM_ROWS=0 forces the k_conv_bf3 path for comparison."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import ops
ops.set_math_mode("bf16x3")
dev = torch.device("cuda:0")
B = 256
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
def ds(H, cin, cout, stride):
    Ho = (H - 1) // stride + 1
    x = torch.randn(B, H, H, cin, device=dev); w = torch.randn(cout, 1, cin, device=dev) * 0.05
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
    out = torch.empty(B, Ho, Ho, cout, device=dev)
    us = t(lambda: ops.conv_image(x, w, sc, sh, out, stride=stride, relu=0))
    mb = 4 * (B * Ho * Ho * (cin + cout)) / 1e6
    print(f"downsample H{H} {cin}->{cout} s{stride}: {us:8.1f} us  {mb / us / 1e3 * 1e3:7.0f} GB/s min-traffic ({mb:.0f} MB)", flush=True)
def dense3(rows, cin, cout):
    a = torch.randn(rows, cin, device=dev); w = torch.randn(cout, 1, cin, device=dev) * 0.05
    sh = torch.randn(cout, device=dev) * 0.1; res = torch.randn(rows, cout, device=dev); out = torch.empty(rows, cout, device=dev)
    us = t(lambda: ops.conv_packed(a, w, None, sh, out, taps=1, m_cap=rows, relu=1, residual2d=res))
    print(f"dense conv3 rows{rows} {cin}->{cout}: {us:8.1f} us  {2.0 * rows * cin * cout / us / 1e6:7.1f} TFLOP/s", flush=True)
ds(56, 64, 256, 1); ds(56, 256, 512, 2); ds(28, 512, 1024, 2); ds(14, 1024, 2048, 2)
dense3(B * 49, 512, 2048)
