#!/usr/bin/env python3
"""Tuning only: ldn_bottleneck_smallmap on the stage-4 shape (bs256, 7x7, 2048 -> 512 -> 2048) at several keep probabilities."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B, H, C, W, gran = int(os.environ.get("SM_B", "256")), 7, 2048, 512, 2
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
colsum = torch.empty(B, 2, C, device=dev)
for keep in [float(a) for a in (sys.argv[1:] or ["0.62", "0.5", "0.75", "1.0"])]:
    gm = (torch.rand(B, W // gran, generator=g) < keep).float().to(dev)
    _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, W // gran, gran, mask_in=gm)
    fn = lambda: ops.bottleneck_smallmap(x, w1s, w2p, w3p, idx, cnt, s1, t1, c1, s2, tab, c2, t3, out, residual=x, colsum=colsum)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"keep {keep}: mean K {cnt.float().mean().item():.0f} max {cnt.max().item()}  {100 * e0.elapsed_time(e1):.1f} us per launch")
