#!/usr/bin/env python3
"""Tuning only: phase cycles of the chained stage-3 launch (k_chain) of the bench model with the LDN_TRACE build.
  hipcc ... -DLDN_TRACE -> tools/ablate/libldn_trace.so (tools/trace_chain.sh builds it);  LDN_LIB_PATH=... python tools/trace_chain.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import laudnet_amd
from laudnet_amd import _lib, ops
import bench
from fill import fill_state_dict, seeded_randn
dev = torch.device("cuda:0")
ops.set_math_mode("bf16x3")
wl = bench.WORKLOADS["channel"]
model = laudnet_amd.uni_resnet101(**dict(wl["kw"], num_classes=1000, input_size=224)).eval()
sd = fill_state_dict(model.state_dict(), 1)
for k in sd:
    if k.endswith("bn3.weight"):
        sd[k] = sd[k] * 0.3
model.load_state_dict(sd); model = model.to(dev)
x = seeded_randn((256, 3, 224, 224), 1000).to(dev).contiguous(memory_format=torch.channels_last)
bench.calibrate_maskers(model, x, float(os.environ.get("KEEP", "0.62")), None)
lib = _lib.load()
trace = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
lib.ldn_debug_set_chain_trace.argtypes = [ctypes.c_void_p]
assert lib.ldn_debug_set_chain_trace(ctypes.c_void_p(trace.data_ptr())) == 0
ld = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
if hasattr(lib, "ldn_debug_set_ld_trace"):
    lib.ldn_debug_set_ld_trace.argtypes = [ctypes.c_void_p]
    assert lib.ldn_debug_set_ld_trace(ctypes.c_void_p(ld.data_ptr())) == 0
with torch.no_grad():
    for _ in range(3):
        model(x, 1.0)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(256, 4).astype(np.float64) / 22.0
tot = t.sum(1)
for i, n in enumerate(["masker", "conv1 (head)", "conv2+conv3 (tail)", "fences"]):
    print(f"{n:20s} mean {t[:, i].mean():9.0f} cycles/block  ({100 * t[:, i].mean() / tot.mean():4.1f} %)   max {t[:, i].max():9.0f}")
print(f"total per block      mean {tot.mean():9.0f}   max image {tot.max():9.0f}  min image {tot.min():9.0f}   (shader-clock cycles, ~2.2 GHz)")

l = ld.cpu().numpy().reshape(256, 8, 8).astype(np.float64) / 22.0
if l.sum() > 0:      # the loader / consumer form (k_chain_ld): cycles per block and wave
    names = ["conv1 loop", "conv1 epilogue", "conv2 loop", "tables+convert", "conv3 loop", "waits: conv1", "conv2", "conv3"]
    for w in (0, 3, 4, 6):
        print(f"consumer wave {w}: " + "  ".join(f"{n} {l[:, w, i].mean():8.0f}" for i, n in enumerate(names)))
    print("loader wave 7:   " + "  ".join(f"{n} {l[:, 7, i].mean():8.0f}" for i, n in ((0, "conv1 stream"), (2, "conv2 stream"), (4, "conv3 stream"), (5, "waits: conv1"), (6, "conv2"), (7, "conv3"))))
print("plan_timeouts (incl. chain hand-off stalls):", ops.plan_timeouts())
