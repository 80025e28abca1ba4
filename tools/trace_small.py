#!/usr/bin/env python3
"""Tuning only: one ldn_bottleneck_smallmap launch with the LDN_TRACE build; per-workgroup phase timestamps.
tools/build_ablate.sh trace_small ldn_small.hip -DLDN_TRACE=1; LDN_LIB_PATH=tools/ablate/libldn_trace_small.so python tools/trace_small.py [keep]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
keep = float(sys.argv[1]) if len(sys.argv) > 1 else 0.62
B, H, C, W, gran = 256, 7, 2048, 512, 2
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, H, C, device=dev).relu_()
w1s = ops.pack_w1_split(torch.randn(W, C, device=dev) * 0.03)
w2p = ops.pack_w2_pairs(torch.randn(W, W, 3, 3, device=dev) * 0.02)
w3p = ops.pack_w3_pairs(torch.randn(C, W, device=dev) * 0.03)
s1, t1, c1 = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1, torch.rand(W, device=dev) * 0.1
s2, c2 = torch.rand(W, device=dev) + 0.5, torch.rand(W, device=dev) * 0.1
tab = torch.randn(16, W, device=dev) * 0.1
t3 = torch.randn(C, device=dev) * 0.1
out = torch.empty_like(x)
gm = (torch.rand(B, W // gran, generator=g) < keep).float().to(dev)
_, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, W // gran, gran, mask_in=gm)
fn = lambda: ops.bottleneck_smallmap(x, w1s, w2p, w3p, idx, cnt, s1, t1, c1, s2, tab, c2, t3, out, residual=x)
lib = _lib.load()
trace = torch.zeros(B * 8 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    fn()
torch.cuda.synchronize()
lib.ldn_debug_set_small_trace.argtypes = [ctypes.c_void_p]
assert lib.ldn_debug_set_small_trace(ctypes.c_void_p(trace.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print("launch us", 1000 * e0.elapsed_time(e1))
t = trace.cpu().numpy().reshape(B, 8, 16).astype(np.float64)
w0 = t[:, 0, :]
tot = (w0[:, 5] - w0[:, 0]).mean()
print("nsub hist", np.bincount(w0[:, 12].astype(int)))
for i, n in enumerate(["conv1 loop", "epilogue 1", "conv2 loop", "epilogue 2", "conv3"]):
    d = w0[:, i + 1] - w0[:, i]
    print(f"{n:12s} mean {d.mean():10.0f} ({100 * d.mean() / tot:4.1f} %)  max {d.max():10.0f}")
print(f"total mean {tot:10.0f}  max {(w0[:, 5] - w0[:, 0]).max():10.0f}  -> cycles per us {tot / (1000 * e0.elapsed_time(e1)):.0f}")
for w in (0, 1, 6, 7):
    tw = t[:, w, :]
    print(f"wave {w}: wait+barrier conv1 {tw[:,6].mean():8.0f} conv2 {tw[:,7].mean():8.0f} conv3 {tw[:,8].mean():8.0f} | DMA issue conv1 {tw[:,9].mean():8.0f} conv2 {tw[:,10].mean():8.0f} conv3 {tw[:,11].mean():8.0f}")
for w in (0, 1, 6, 7):
    tw = t[:, w, :]
    print(f"wave {w}: conv3 fast loop: vmcnt wait {tw[:,13].mean():8.0f}  barrier {tw[:,14].mean():8.0f}  DMA issue {tw[:,15].mean():8.0f}")
