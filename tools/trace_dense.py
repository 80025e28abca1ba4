#!/usr/bin/env python3
"""Tuning only: cycle split of k_dense's K loop (LDN_TRACE build): wait(vmcnt) / barrier / DMA issue / B-operand prep / MFMA steps.
LDN_LIB_PATH=tools/ablate/libldn_trace.so python tools/trace_dense.py [M K N]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import _lib, ops
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50176, 512, 1024)
dev = torch.device("cuda:0")
ops.set_math_mode("bf16x3")
a = torch.randn(M, K, device=dev)
w = torch.randn(N, 1, K, device=dev) * 0.05
sh = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev)
fn = lambda: ops.conv_rows(a, w, None, sh, out, taps=1, m_cap=M, relu=1)
for _ in range(3):
    fn()
torch.cuda.synchronize()
lib = _lib.load()
nwg = ((M + 255) // 256 + 7) // 8 * 8 * ((N + 255) // 256) * 2 + 64
trace = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
lib.ldn_debug_set_dense_trace.argtypes = [ctypes.c_void_p]
assert lib.ldn_debug_set_dense_trace(ctypes.c_void_p(trace.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record(); torch.cuda.synchronize()
print(f"M {M} K {K} N {N}: launch {200 * e0.elapsed_time(e1):.1f} us")
t = trace.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)
t = t[t[:, 0, 7] > 0]
w0 = t[:, 0, :]
nch = w0[0, 7]
print(f"workgroups traced {len(t)}, chunks {nch:.0f}")
for i, n in enumerate(["vmcnt wait", "barrier", "DMA issue", "B prep (reads + split)", "MFMA steps"]):
    print(f"{n:24s} {w0[:, i].mean() / nch:8.0f} cycles per chunk")
print(f"K loop total {w0[:, 5].mean() / nch:8.0f} cycles per chunk; epilogue {w0[:, 6].mean():8.0f} cycles")
