#!/usr/bin/env python3
"""Tuning only: run one stage-3 conv2 launch with the LDN_TRACE build and analyse per-block timestamps.
LDN_LIB_PATH=tools/ablate/libldn_trace.so python tools/trace_conv.py [kind] [stage]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import _lib, ops  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "conv2"
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, Cin, W = {1: (56, 256, 64), 2: (28, 512, 128), 3: (14, 1024, 256), 4: (7, 2048, 512)}[stage]
dev = torch.device("cuda:0")
B, gran = 256, 2
g = torch.Generator().manual_seed(0)
G = W // gran
gm = (torch.rand(B, G, generator=g) < 0.62).float().to(dev)
_, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm)
x = torch.randn(B, H, H, Cin, device=dev)
h1 = torch.randn(B, H, H, W, device=dev)
h2 = torch.randn(B, H, H, W, device=dev)
out = torch.empty(B, H, H, Cin, device=dev)
w1 = torch.randn(W, 1, Cin, device=dev) * 0.05
w2 = torch.randn(9, W, W, device=dev) * 0.05      # k-major [taps][cin][cout]
w3 = torch.randn(1, W, Cin, device=dev) * 0.05    # k-major
sW, tW = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1
sC, tC = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
tab = torch.randn(16, W, device=dev) * 0.1
cW = torch.rand(W, device=dev) * 0.1
fns = {
    "conv1": lambda: ops.conv_image(x, w1, sW, tW, h1, n_idx=idx, n_cnt=cnt, post_sub=cW, relu=1),
    "conv2": lambda: ops.conv_image(h1, w2, sW, tab, h2, ksize=3, stride=1, k_idx=idx, k_cnt=cnt, kgran=gran, n_idx=idx,
                                    n_cnt=cnt, post_sub=cW, relu=1),
    "conv3": lambda: ops.conv_image(h2, w3, None, tC, out, k_idx=idx, k_cnt=cnt, kgran=gran, relu=1, residual=x),
}
fn = fns[kind]
lib = _lib.load()
nblk = 1 << 17
trace = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
lib.ldn_debug_set_trace.argtypes = [ctypes.c_void_p]
for _ in range(3):
    fn()
torch.cuda.synchronize()
assert lib.ldn_debug_set_trace(ctypes.c_void_p(trace.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record()
torch.cuda.synchronize()
print("launch us", 1e3 * e0.elapsed_time(e1))
t = trace.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
prod = t[t[:, 2] == 0]
t = t[t[:, 2] != 0]
if len(prod):
    pn = prod[:, 4] & 0xffffffff; big = pn == pn.max()
    print('PRODUCER wave (largest blocks): dur', (prod[big, 1] - prod[big, 0]).mean(), ' wait-own-loads', ((prod[big, 5] >> 32) & 0xffffffff).mean(), ' wait-barrier', (prod[big, 4] >> 32).mean(), ' issue', (prod[big, 5] & 0xffffffff).mean())
t0, t1, hw, xcc, nt_bar, mma_iss, t_pro, t_epi = [t[:, i] for i in range(8)]
ntiles = nt_bar & 0xffffffff; t_bar = nt_bar >> 32; t_mma = (mma_iss >> 32) & 0xffffffff; t_iss = mma_iss & 0xffffffff; tm = t1
base = t0.min()
print("blocks traced", len(t), " memtime span", (t1.max() - base), "ticks")
dur = t1 - t0
print("ticks/us estimate:", (t1.max() - base) / (1e3 * e0.elapsed_time(e1)))
for nt_ in sorted(set(ntiles.tolist())):
    m = ntiles == nt_
    print(f" ntiles={nt_:3d}: n={m.sum():5d}  dur mean {dur[m].mean():9.0f} min {dur[m].min():8d} max {dur[m].max():8d}  start mean {(t0[m]-base).mean():9.0f} max {(t0[m]-base).max():9d}  wave0: barrier-wait {t_bar[m].mean():9.0f} issue {t_iss[m].mean():8.0f} mma {t_mma[m].mean():9.0f} prologue {t_pro[m].mean():8.0f} epilogue {t_epi[m].mean():8.0f}")
cu = ((xcc & 0xf) << 16) | (hw & 0xff00) | ((hw >> 13) & 7) << 4 | ((hw >> 12) & 1)
cu_key = (xcc & 0xf) * 10000 + ((hw >> 13) & 7) * 1000 + ((hw >> 12) & 1) * 100 + ((hw >> 8) & 0xf)
uk = np.unique(cu_key)
print("distinct CUs seen:", len(uk)); big = ntiles == ntiles.max(); print("shader clock GHz (memtime/realtime@100MHz) on largest blocks:", (dur[big] / np.maximum(xcc[big],1) * 0.1).mean())
# concurrency per CU: sweep
conc = []
busy = []
for k in uk[:]:
    m = cu_key == k
    ev = sorted([(a, 1) for a in t0[m]] + [(b_, -1) for b_ in t1[m]])
    c = 0; last = ev[0][0]; area = 0; active = 0
    for tt, d in ev:
        area += c * (tt - last); active += (tt - last) if c > 0 else 0
        last = tt; c += d
    conc.append(area / max(active, 1)); busy.append(active)
print("mean concurrency while busy per CU:", np.mean(conc), " mean busy ticks per CU:", np.mean(busy), "max", np.max(busy), "min", np.min(busy))
print("blocks per CU: mean", len(t) / len(uk), " max", max((cu_key == k).sum() for k in uk))
