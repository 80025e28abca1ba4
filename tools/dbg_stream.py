#!/usr/bin/env python3
"""Tuning/debug: k_conv1x1_stream against an fp64 reference on small dense / gathered cases."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd import ops
ops.set_math_mode("bf16x3")
dev = torch.device("cuda:0")
torch.manual_seed(0)

def report(name, got, want):
    err = (got.double() - want).abs()
    bad = err > 1e-3 * (1 + want.abs())
    print(f"{name}: max err {err.max().item():.3e}  bad {int(bad.sum())}/{bad.numel()}", flush=True)
    if bad.any():
        idx = bad.nonzero()
        print("   first bad", idx[:5].tolist(), " rows bad:", sorted(set(idx[:, -2].tolist()))[:20], " cols bad (first 20):", sorted(set(idx[:, -1].tolist()))[:20])

def dense(B, H, cin, cout, stride=1, resid=True):
    Ho = (H - 1) // stride + 1
    x = torch.randn(B, H, H, cin, device=dev)
    w = torch.randn(cout, 1, cin, device=dev) * (2.0 / cin) ** 0.5
    sh = torch.randn(cout, device=dev) * 0.1
    res = torch.randn(B, Ho, Ho, cout, device=dev) if resid else None
    out = torch.full((B, Ho, Ho, cout), float("nan"), device=dev)
    ops.conv_image(x, w, None, sh, out, ksize=1, stride=stride, relu=1, residual=res)
    torch.cuda.synchronize()
    xs = x[:, ::stride, ::stride].double()
    want = xs @ w[:, 0].double().T + sh.double()
    if resid: want = want + res.double()
    want = torch.relu(want)
    report(f"dense B{B} H{H} cin{cin} cout{cout} s{stride} res{resid}", out.reshape(-1, Ho * Ho, cout), want.reshape(-1, Ho * Ho, cout))

def gathered(B, H, W, cout, p=0.6, gran=2):
    G = W // gran
    gm = (torch.rand(B, G, device=dev) < p).float()
    _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm)
    h2 = torch.randn(B, H, H, W, device=dev)
    # left-packed: zero the columns cnt..roundup4
    for b in range(B):
        n = int(cnt[b]); h2[b, :, :, n:] = float("nan"); h2[b, :, :, n:(n + 3) // 4 * 4] = 0
    w = torch.randn(1, W, cout, device=dev) * (2.0 / W) ** 0.5
    sh = torch.randn(cout, device=dev) * 0.1
    res = torch.randn(B, H, H, cout, device=dev)
    out = torch.full((B, H, H, cout), float("nan"), device=dev)
    ops.conv_image(h2, w, None, sh, out, k_idx=idx, k_cnt=cnt, kgran=gran, relu=1, residual=res)
    torch.cuda.synchronize()
    want = torch.empty(B, H, H, cout, dtype=torch.float64, device=dev)
    for b in range(B):
        n = int(cnt[b]); ch = idx[b, :n].long()
        want[b] = torch.relu(h2[b, :, :, :n].double() @ w[0, ch].double() + sh.double() + res[b].double())
    report(f"gathered B{B} H{H} W{W} cout{cout}", out.reshape(B, H * H, cout), want.reshape(B, H * H, cout))

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "dense"):
    dense(1, 8, 32, 256)
    dense(2, 14, 64, 256)
    dense(2, 14, 64, 256, resid=False)
    dense(3, 28, 128, 512, stride=2)
    dense(2, 40, 96, 384)
if which in ("all", "gather"):
    gathered(2, 8, 64, 256)
    gathered(3, 14, 256, 1024)
    gathered(2, 56, 64, 256)
print("done")
