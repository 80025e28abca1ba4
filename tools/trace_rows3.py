#!/usr/bin/env python3
"""Tuning only: cycle split of k_rows3's K loop (LDN_TRACE build): vmcnt wait / barrier / body per K64 step, per wave.
tools/build_ablate.sh trace3 ldn_rows3.hip -DLDN_TRACE; LDN_LIB_PATH=tools/ablate/libldn_trace3.so python tools/trace_rows3.py [stage] [p]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from laudnet_amd import _lib, ops
from fill import seeded_bernoulli
st = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
H, Cin, W, S = {1: (56, 256, 64, 14), 2: (28, 512, 128, 7), 3: (14, 1024, 256, 7)}[st]
B = 256
dev = torch.device("cuda:0")
ops.set_math_mode("bf16x3")
ix = ops.mask_to_index(seeded_bernoulli((B, S, S), p, 5).to(dev), H, H, 1)
n3 = int(ix.cnt[0])
h1 = ops.presplit_rows(torch.relu(torch.randn(ix.cap1, W, device=dev)))
w2 = torch.randn(W, 9, W, device=dev) * 0.05
sW, tW = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1
h2 = torch.empty(ix.cap3, W, device=dev)
fn = lambda: ops.conv3x3_rows_ps(h1, ix.nbr, w2, sW, tW, h2, m_count=ix.cnt[0:1], m_cap=ix.cap3, out_presplit=True, rows_hint=n3)
for _ in range(3):
    fn()
torch.cuda.synchronize()
lib = _lib.load()
nwg = (ix.cap3 // 256 + 16) * 8
trace = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
lib.ldn_debug_set_rows3_trace.argtypes = [ctypes.c_void_p]
assert lib.ldn_debug_set_rows3_trace(ctypes.c_void_p(trace.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record(); torch.cuda.synchronize()
print(f"stage {st} rows {n3}: launch {200 * e0.elapsed_time(e1):.1f} us")
t = trace.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)
t = t[t[:, 0, 7] > 0]
ns = t[0, 0, 7]
print(f"workgroups traced {len(t)}, K64 steps {ns:.0f}; MFMA floor per step and SIMD: {2 * 2 * 2 * (W // 32 if W < 128 else 4) * 3 * 32} cycles")
for wv in (0, 4, 7):
    w = t[:, wv, :]
    print(f"wave {wv}: vmcnt wait {w[:, 0].mean() / ns:7.0f}  barrier {w[:, 1].mean() / ns:7.0f}  body {w[:, 2].mean() / ns:7.0f}  per K64 step;"
          f"  loop {w[:, 5].mean():9.0f}  epilogue {w[:, 6].mean():8.0f} cycles")
