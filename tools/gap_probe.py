#!/usr/bin/env python3
"""Tuning: which kernels leave a gap behind them?  Launches each candidate followed by a tiny kernel, ten times, on one stream; run it under
`rocprofv3 --kernel-trace` and read the gaps with tools/gap_probe.py --parse <db>.  (the ~5.6 us behind k_dense / k_rows3 / k_plan.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def parse(db_path):
    import collections
    import sqlite3
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end, 0 from kernels order by start").fetchall()
    if "--seq" in sys.argv:     # every transition in order, from the first marker (FillFunctor) on
        on = False
        for i in range(len(rows) - 1):
            if "FillFunctor" in rows[i][0]:
                on = True
                print("---- marker")
            if on and "distribution" not in rows[i][0]:
                print(f"{rows[i][0].replace('void ', '').replace('ldn::', '')[:60]:62s} dur {(rows[i][2] - rows[i][1]) / 1e3:8.1f}  gap behind {(rows[i + 1][1] - rows[i][2]) / 1e3:8.2f}")
            if on and "ran" == 0:
                break
        return
    gaps = collections.defaultdict(list)
    for i in range(len(rows) - 1):
        n = rows[i][0].replace("void ", "").replace("ldn::", "")[:70] + f" grid {rows[i][3]}"
        gaps[n].append(((rows[i + 1][1] - rows[i][2]) / 1e3, (rows[i][2] - rows[i][1]) / 1e3))
    print(f"{'kernel':100s} {'n':>4s} {'dur us':>8s} {'gap behind: median':>20s} {'min':>7s} {'max':>7s}")
    for n, g in gaps.items():
        gs = sorted(v[0] for v in g)
        print(f"{n:100s} {len(g):4d} {sum(v[1] for v in g) / len(g):8.1f} {gs[len(gs) // 2]:20.2f} {gs[0]:7.2f} {gs[-1]:7.2f}")


def main():
    import torch
    from laudnet_amd import ops
    from fill import seeded_bernoulli
    dev = torch.device("cuda:0")
    ops.set_math_mode("bf16x3")
    B, H, Cin, W, S = 256, 14, 1024, 256, 7
    ix = ops.mask_to_index(seeded_bernoulli((B, S, S), 0.5, 5).to(dev), H, H, 1)
    n3, n1 = int(ix.cnt[0]), int(ix.cnt[1])
    x = torch.relu(torch.randn(B * H * H, Cin, device=dev))
    x4 = x.view(B, H, H, Cin)
    w1 = torch.randn(W, 1, Cin, device=dev) * 0.05
    w2 = torch.randn(W, 9, W, device=dev) * 0.05
    w3 = torch.randn(Cin, 1, W, device=dev) * 0.05
    sW, tW = torch.rand(W, device=dev) + 0.5, torch.randn(W, device=dev) * 0.1
    tC = torch.randn(Cin, device=dev) * 0.1
    h1 = torch.empty(ix.cap1, W, device=dev)
    h2 = torch.empty(ix.cap3, W, device=dev)
    out = x.clone()
    tiny = torch.zeros(64, device=dev)
    mw, mb = torch.randn(2, Cin, device=dev), torch.zeros(2, device=dev)
    nbr_exact = ix.nbr.view(-1, 9)[: (n3 // 2048) * 2048].contiguous()       # a multiple of 8 M tiles: no workgroup exits early
    a1 = torch.randn(4096, 4096, device=dev)
    cands = {
        "rows3, device count (early exits)": lambda: ops.conv3x3_rows_ps(h1, ix.nbr, w2, sW, tW, h2, m_count=ix.cnt[0:1], m_cap=ix.cap3, out_presplit=True, rows_hint=n3),
        "rows3, exact rows (no early exit)": lambda: ops.conv3x3_rows_ps(h1, nbr_exact, w2, sW, tW, h2, m_cap=nbr_exact.shape[0], out_presplit=True),
        "conv1 (k_dense OF)": lambda: ops.conv_rows_ps(x, w1, sW, tW, h1, out_presplit=True, a_rows=ix.idx1, m_count=ix.cnt[1:2], m_cap=ix.cap1, rows_hint=n1),
        "conv3 (k_dense PS)": lambda: ops.conv_rows_ps(h2, w3, None, tC, out, a_presplit=True, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_rows=ix.idx3, residual2d=out, rows_hint=n3),
        "spatial masker": lambda: ops.spatial_masker(x4, mw, mb, 1, S),
        "mask_to_index (k_plan)": lambda: ops.mask_to_index(seeded_patch, H, H, 1),
        "torch matmul 4096": lambda: torch.matmul(a1, a1),
        "torch relu 205 MB": lambda: torch.relu_(x),
    }
    global seeded_patch
    seeded_patch = seeded_bernoulli((B, S, S), 0.5, 5).to(dev)
    c1, c2, c3 = cands["conv1 (k_dense OF)"], cands["rows3, device count (early exits)"], cands["conv3 (k_dense PS)"]
    seqs = {"SEQ rows3 x10 back to back": [c2] * 10, "SEQ conv1 rows3 conv3 x4": [c1, c2, c3] * 4,
            "SEQ conv1 tiny rows3 tiny conv3 tiny x4": [c1, None, c2, None, c3, None] * 4,
            "SEQ masker x6": [cands["spatial masker"]] * 6, "SEQ matmul x4": [cands["torch matmul 4096"]] * 4}
    from laudnet_amd.laud_resnet import Bottleneck
    from fill import fill_state_dict
    blk = Bottleneck(1024, 256, stride=1, dyn_mode="spatial", output_size=14, mask_spatial_granularity=2).eval()
    blk.load_state_dict(fill_state_dict(blk.state_dict(), 41))
    blk = blk.to(dev)
    xb = torch.relu(torch.randn(B, 1024, 14, 14, device=dev)).contiguous(memory_format=torch.channels_last)
    forced = seeded_bernoulli((B, 1, 7, 7), 0.5, 43).to(dev)

    def run_blk():
        with torch.no_grad():
            blk.run_dynamic(xb, inplace=True)
    run_blk(); run_blk()
    seqs["SEQ real block (own masker) x5"] = [run_blk] * 5

    def run_forced():
        blk.forced_spatial_mask = forced
        run_blk()
        blk.forced_spatial_mask = None
    seqs["SEQ real block (forced mask) x5"] = [run_forced] * 5
    for name, seq in seqs.items():
        torch.cuda.synchronize()
        tiny.fill_(7.0)      # marker in the trace
        for fn in seq:
            if fn is None:
                tiny.add_(1.0)
            else:
                fn()
        torch.cuda.synchronize()
        print("ran", name, flush=True)
    for name, fn in cands.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        for _ in range(10):
            fn()
            tiny.add_(1.0)
        torch.cuda.synchronize()
        print("ran", name, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        main()
