#!/usr/bin/env python3
"""Tuning only: one ldn_bottleneck_tail launch with the LDN_TRACE build; per-workgroup phase timestamps.
LDN_LIB_PATH=tools/ablate/libldn_trace.so python tools/trace_tail.py [stage]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import _lib, ops  # noqa: E402
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, Cin, W = {1: (56, 256, 64), 2: (28, 512, 128), 3: (14, 1024, 256)}[stage]
dev = torch.device("cuda:0")
B, gran = 256, 2
g = torch.Generator().manual_seed(0)
G = W // gran
gm = (torch.rand(B, G, generator=g) < 0.62).float().to(dev)
_, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm)
h1 = (torch.randn(B, H, H, W, device=dev) * 0.1)
x = torch.randn(B, H, H, Cin, device=dev)
out = torch.empty_like(x)
w2p = ops.pack_w2_pairs(torch.randn(W, W, 3, 3, device=dev) * 0.05)
w3p = ops.pack_w3_pairs(torch.randn(Cin, W, device=dev) * 0.05)
sW, cW = torch.rand(W, device=dev) + 0.5, torch.rand(W, device=dev) * 0.1
tab = torch.randn(16, W, device=dev) * 0.1
tC = torch.randn(Cin, device=dev) * 0.1
fn = lambda: ops.bottleneck_tail(h1, w2p, w3p, idx, cnt, sW, tab, cW, tC, out, residual=x)
lib = _lib.load()
nwg = B * ops.bottleneck_tail_splits(H, H, W) // 8
trace = torch.zeros(nwg * 8 * 12, dtype=torch.int64, device=dev)
for _ in range(3):
    fn()
torch.cuda.synchronize()
has_trace = hasattr(lib, "ldn_debug_set_tail_trace")
if has_trace:
    lib.ldn_debug_set_tail_trace.argtypes = [ctypes.c_void_p]
    assert lib.ldn_debug_set_tail_trace(ctypes.c_void_p(trace.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"stage {stage}: launch us", 200 * e0.elapsed_time(e1), " workgroups", nwg)
trace.zero_(); fn(); torch.cuda.synchronize()
if not has_trace:
    sys.exit(0)
t = trace.cpu().numpy().reshape(nwg, 8, 12).astype(np.float64)
w0 = t[:, 0, :]
print("nsub histogram", np.bincount(w0[:, 10].astype(int)))
names = ["setup", "conv2 loop", "convert", "conv3 loop"]
for i, n in enumerate(names):
    print(f"{n:12s} mean {np.mean(w0[:, i + 1] - w0[:, i]):10.0f} cycles   max {np.max(w0[:, i + 1] - w0[:, i]):10.0f}")
print(f"total        mean {np.mean(w0[:, 4] - w0[:, 0]):10.0f}   max {np.max(w0[:, 4] - w0[:, 0]):10.0f}")
for w in (0, 3, 6, 7):
    tw = t[:, w, :]
    print(f"wave {w}: conv2 wait+barrier {tw[:,5].mean():9.0f} | conv3: K loops {tw[:,6].mean():9.0f}  vmcnt wait {tw[:,7].mean():9.0f}  epilogue {tw[:,8].mean():9.0f}  barrier {tw[:,9].mean():9.0f}")


