#!/usr/bin/env python3
"""Tuning only: the fused stem alone (bs256, 224^2) under the library named by LDN_LIB_PATH; prints us per launch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from laudnet_amd import ops, load_library
load_library()
dev = "cuda:0"
w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
frag = ops.pack_stem_weights(w)
shift = torch.zeros(64, device=dev)
x = torch.randn(256, 224, 224, 3, device=dev)
for _ in range(3):
    ops.stem_conv_pool(x, frag, shift, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.stem_conv_pool(x, frag, shift, 64)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("LDN_LIB_PATH", "release"), "us per launch: %.1f" % (100 * e0.elapsed_time(e1)))
