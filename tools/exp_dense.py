#!/usr/bin/env python3
"""Tuning: what bounds k_dense's K loop?  Times one shared-weight 1x1 (stage-3 projection shape by default) with (a) its real
activation rows, (b) every M tile reading the SAME 256 rows (activations L1/L2-resident: memory system out of the picture).
usage: python tools/exp_dense.py [M K N]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import ops
ops.set_math_mode("bf16x3")
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50176, 512, 1024)
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev)
w = torch.randn(N, 1, K, device=dev) * 0.05
sh = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev)
same = (torch.arange(M, device=dev) % 256).to(torch.int32)
ident = torch.arange(M, device=dev).to(torch.int32)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
fl = 2.0 * M * K * N
for name, rows in (("real rows", None), ("identity list", ident), ("same 256 rows", same)):
    us = t(lambda: ops.conv_rows(a, w, None, sh, out, a_rows=rows, taps=1, m_cap=M, relu=1))
    print(f"LDN_DENSE16={os.environ.get('LDN_DENSE16', '-')} M {M} K {K} N {N} [{name:14s}]: {us:7.1f} us  {3 * fl / us / 1e6:7.1f} TFLOP/s executed", flush=True)
