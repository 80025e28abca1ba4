import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
from laudnet_amd import ops
from fill import seeded_bernoulli, seeded_randn
DEV = "cuda:0"
ops.set_math_mode("bf16x3")
for (B, H, stride, C, cout, p) in [(3, 14, 1, 64, 64, 0.5), (2, 14, 1, 256, 256, 0.5), (5, 7, 1, 128, 64, 1.0), (40, 14, 1, 256, 256, 0.5)]:
    Ho = H // stride
    patch = seeded_bernoulli((B, Ho, Ho), p, 11)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    n3 = int(ix.cnt[0])
    h1 = torch.relu(seeded_randn((ix.cap1, C), 12)).to(DEV)
    w = (seeded_randn((cout, 9, C), 13) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    g = torch.Generator().manual_seed(14)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * 0.1).to(DEV)
    nbr = ix.nbr.view(-1, 9)[:n3].long()
    h1z = torch.cat((h1.double(), torch.zeros(1, C, dtype=torch.float64, device=DEV)))
    gath = h1z[torch.where(nbr >= 0, nbr, torch.full_like(nbr, ix.cap1))]
    ref = torch.relu(torch.einsum("mtk,ntk->mn", gath, w.double()) * sc.double() + sh.double())
    for rep in range(4):
        want = torch.zeros(ix.cap3, cout, device=DEV)
        ops.conv_rows(h1, w, sc, sh, want, a_rows=ix.nbr, taps=9, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1)
        got = torch.zeros(ix.cap3, cout, device=DEV)
        ops.conv3x3_rows_ps(ops.presplit_rows(h1), ix.nbr, w, sc, sh, got, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_presplit=False)
        torch.cuda.synchronize()
        d = (got[:n3] - want[:n3]).abs()
        print(f"C {C} cout {cout} n3 {n3} rep {rep}: mismatches {(d > 0).sum().item()} max {d.max().item():.3e} | rows3 vs fp64 {(got[:n3].double() - ref).abs().max().item():.3e}  dense vs fp64 {(want[:n3].double() - ref).abs().max().item():.3e}"
              f" bad rows {torch.nonzero(d.amax(dim=1) > 0).flatten()[:8].tolist()}")
