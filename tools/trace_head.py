#!/usr/bin/env python3
"""Tuning only: one ldn_bottleneck_head launch with the LDN_TRACE build; per-wave cycle split of the K loop.
LDN_LIB_PATH=tools/ablate/libldn_trace.so python tools/trace_head.py [stage]"""
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
G = W // gran
gm = (torch.rand(B, G, generator=torch.Generator().manual_seed(0)) < 0.62).float().to(dev)
_, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm)
x = torch.randn(B, H, H, Cin, device=dev)
h1 = torch.empty(B, H, H, W, device=dev)
w1s = ops.pack_w1_split(torch.randn(W, Cin, device=dev) * 0.05)
s1, t1, c1 = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1, torch.rand(W, device=dev) * 0.1
fn = lambda: ops.bottleneck_head(x, w1s, idx, cnt, s1, t1, c1, h1)
lib = _lib.load()
nwg = B * ((H * H + 255) // 256)
trace = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    fn()
torch.cuda.synchronize()
lib.ldn_debug_set_head_trace.argtypes = [ctypes.c_void_p]
assert lib.ldn_debug_set_head_trace(ctypes.c_void_p(trace.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record(); torch.cuda.synchronize()
print(f"stage {stage}: launch us", 200 * e0.elapsed_time(e1), " workgroups", nwg)
trace.zero_(); fn(); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(nwg, 8, 8).astype(np.float64)
names = ["vmcnt wait", "barrier", "DMA issue", "x read+split", "MFMA steps", "K loop total"]
for w in (0, 3, 5, 6, 7):
    tw = t[:, w, :]
    ch = max(tw[:, 6].mean(), 1)
    print(f"wave {w}: " + "  ".join(f"{n} {tw[:, i].mean() / ch:7.0f}" for i, n in enumerate(names)) + f"   (cycles per chunk; {ch:.0f} chunks, nsub {tw[:,7].mean():.2f})")
