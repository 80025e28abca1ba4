#!/usr/bin/env python3
"""Tuning only: phase split of k_dense's EPILOGUE per 32-column subtile (build: tools/build_ablate.sh epi ldn_dense.hip -DLDN_TRACE -DLDN_TRACE_EPI):
loads issued + LDS transpose written / loads ready (forced before the first store) / affine + stores issued / LDS drain.
LDN_LIB_PATH=tools/ablate/libldn_epi.so python tools/trace_dense_epi.py [M K N [residual]]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import _lib, ops
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (25000, 256, 1024)
resid = len(sys.argv) > 4
dev = torch.device("cuda:0")
ops.set_math_mode("bf16x3")
a = torch.randn(M, K, device=dev)
w = torch.randn(N, 1, K, device=dev) * 0.05
sh = torch.zeros(N, device=dev)
out = torch.zeros(M, N, device=dev)
fn = lambda: ops.conv_rows(a, w, None, sh, out, taps=1, m_cap=M, relu=1, residual2d=out if resid else None)
for _ in range(3):
    fn()
torch.cuda.synchronize()
lib = _lib.load()
nwg = ((M + 255) // 256 + 7) // 8 * 8 * ((N + 63) // 64) * 2 + 64
trace = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
lib.ldn_debug_set_dense_trace.argtypes = [ctypes.c_void_p]
assert lib.ldn_debug_set_dense_trace(ctypes.c_void_p(trace.data_ptr())) == 0
fn(); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)
t = t[t[:, 0, 5] > 0]
w0 = t[:, 0, :]
print(f"M {M} K {K} N {N} residual {resid}: workgroups traced {len(t)}; epilogue total {w0[:, 6].mean():.0f} cycles")
for i, n in enumerate(["loads issued + transpose written", "loads ready (before first store)", "affine + stores issued", "LDS drain + wave barrier"]):
    print(f"  {n:36s} {w0[:, i].mean():8.0f} cycles per tile")
