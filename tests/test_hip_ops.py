"""GPU parity tests of the C-ABI ops (through laudnet_amd.ops -> libldn_hip.so) against the oracle.
Integer/index work must be bit-exact; fp32 work within the tolerance written at each assert."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fill import seeded_bernoulli, seeded_randn
from oracle import index_ref as IR
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _set_math_mode(math_mode):
    from laudnet_amd import ops as _ops
    _ops.set_math_mode(math_mode)
    yield
    _ops.set_math_mode("fp32")


@pytest.fixture(scope="module")
def ops():
    from laudnet_amd import ops as _ops, load_library
    load_library()  # raises if libldn_hip.so is missing -- no fallback
    return _ops


# ------------------------------------------------------------------ index work (bit-exact)
@pytest.mark.parametrize("B,S,Ho,stride,p", [
    (3, 14, 14, 1, 0.5), (3, 3, 14, 1, 0.5), (2, 3, 14, 2, 0.4), (4, 1, 7, 1, 0.5), (4, 1, 7, 2, 0.5),
    (2, 7, 28, 2, 0.5), (5, 14, 56, 1, 0.3), (2, 9, 28, 1, 0.5), (2, 56, 56, 1, 0.5), (8, 7, 7, 2, 0.0),
    (8, 7, 7, 2, 1.0), (1, 14, 56, 2, 0.5), (300, 2, 7, 1, 0.5),
])
def test_mask_to_index(ops, B, S, Ho, stride, p):
    patch = seeded_bernoulli((B, S, S), p, 7 + B + S + Ho)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    torch.cuda.synchronize()
    m3 = IR.upsample_patch_mask(patch.numpy() > 0.5, Ho)
    m1 = IR.dilate_mask(m3, stride, 1)
    idx3, pre3 = IR.nonzero_rows(m3)
    idx1, pre1 = IR.nonzero_rows(m1)
    cnt = ix.cnt.cpu().numpy()
    assert cnt[0] == len(idx3) and cnt[1] == len(idx1)
    assert np.array_equal(ix.idx3.cpu().numpy()[:cnt[0]], idx3)
    assert np.array_equal(ix.idx1.cpu().numpy()[:cnt[1]], idx1)
    assert np.array_equal(ix.pre3.cpu().numpy(), pre3) and np.array_equal(ix.pre1.cpu().numpy(), pre1)
    assert np.array_equal(ix.pos3.cpu().numpy(), IR.position_map(m3).reshape(-1))
    assert np.array_equal(ix.pos1.cpu().numpy(), IR.position_map(m1).reshape(-1))
    nbr = ix.nbr.cpu().numpy().reshape(-1, 9)[:cnt[0]]
    assert np.array_equal(nbr, IR.neighbour_table(m3, m1, stride))
    stats = ix.stats.cpu().numpy()
    want = np.array([patch.mean().item(), m3.mean(), m1.mean()], dtype=np.float32)
    assert np.allclose(stats, want, atol=1e-6)


def _check_index(ops, patch, Ho, Wo, stride):
    ix = ops.mask_to_index(patch.to(DEV), Ho, Wo, stride)
    torch.cuda.synchronize()
    m3 = IR.upsample_patch_mask(patch.numpy() > 0.5, Ho, Wo)
    m1 = IR.dilate_mask(m3, stride, 1)
    idx3, pre3 = IR.nonzero_rows(m3)
    idx1, pre1 = IR.nonzero_rows(m1)
    cnt = ix.cnt.cpu().numpy()
    assert cnt[0] == len(idx3) and cnt[1] == len(idx1)
    assert np.array_equal(ix.idx3.cpu().numpy()[:cnt[0]], idx3)
    assert np.array_equal(ix.idx1.cpu().numpy()[:cnt[1]], idx1)
    assert np.array_equal(ix.pre3.cpu().numpy(), pre3) and np.array_equal(ix.pre1.cpu().numpy(), pre1)
    assert np.array_equal(ix.pos3.cpu().numpy(), IR.position_map(m3).reshape(-1))
    assert np.array_equal(ix.pos1.cpu().numpy(), IR.position_map(m1).reshape(-1))
    nbr = ix.nbr.cpu().numpy().reshape(-1, 9)[:cnt[0]]
    assert np.array_equal(nbr, IR.neighbour_table(m3, m1, stride))
    want = np.array([patch.mean().item(), m3.mean(), m1.mean()], dtype=np.float32)
    assert np.allclose(ix.stats.cpu().numpy(), want, atol=1e-6)


@pytest.mark.parametrize("B,S,Ho,Wo,stride,p", [
    (2, 1, 100, 152, 2, 0.5),      # detection-size layer skip (800x1216 input, stage 2): 200x304 input map, bands of rows
    (2, 1, 200, 304, 1, 0.5),      # ... stage 1
    (1, 25, 200, 304, 1, 0.3),     # 8x8 patches on a map that does not fit one workgroup's LDS
    (3, 7, 120, 120, 1, 0.5), (2, 5, 96, 160, 2, 0.4), (1, 3, 14, 40, 1, 0.5), (2, 4, 9, 23, 2, 0.5),
])
def test_mask_to_index_nonsquare_and_large(ops, B, S, Ho, Wo, stride, p):
    """Non-square maps and maps whose per-image tables exceed one workgroup's LDS (the banded build): bit-exact lists."""
    _check_index(ops, seeded_bernoulli((B, S, S), p, 11 + B + S + Ho), Ho, Wo, stride)


@pytest.mark.parametrize("B,S,Ho,Wo,stride,p", [
    (3, 14, 14, 14, 1, 0.5), (2, 3, 14, 14, 2, 0.4), (4, 1, 7, 7, 2, 0.5), (2, 7, 28, 28, 2, 0.5), (5, 14, 56, 56, 1, 0.3),
    (2, 56, 56, 56, 1, 0.5), (8, 7, 7, 7, 2, 0.0), (8, 7, 7, 7, 2, 1.0), (1, 14, 56, 56, 2, 0.5), (2, 9, 28, 20, 1, 0.5),
])
def test_mask_to_index_forced_bands(ops, B, S, Ho, Wo, stride, p, monkeypatch):
    """The banded build forced onto small maps with a tiny LDS budget (one or two rows per band): same lists as the whole-image
    kernel and the oracle -- every band boundary is exercised (halo rows of mask3, the neighbour rows' positions)."""
    monkeypatch.setenv("LDN_INDEX_BANDS", "1")
    for budget in ("1", "6000"):
        monkeypatch.setenv("LDN_INDEX_BAND_LDS", budget)
        _check_index(ops, seeded_bernoulli((B, S, S), p, 7 + B + S + Ho), Ho, Wo, stride)


@pytest.mark.parametrize("B,Ho,Wo,stride,p", [(9, 14, 14, 1, 0.5), (256, 14, 14, 1, 0.5), (7, 28, 28, 2, 0.4), (5, 56, 56, 1, 0.6),
                                               (3, 56, 56, 2, 0.5), (8, 7, 7, 2, 0.0), (8, 7, 7, 1, 1.0), (4, 20, 33, 2, 0.5),
                                               (1, 14, 14, 1, 1.0), (2, 100, 168, 1, 0.5)])
def test_mask_to_index_whole_image_masks(ops, B, Ho, Wo, stride, p, monkeypatch):
    """One decision per image (layer skip, S = 1): the closed-form kernel (k_layer_index) against the oracle and, list by list,
    against the general kernels (LDN_INDEX_GENERIC=1)."""
    patch = seeded_bernoulli((B, 1, 1), p, 5 + B + Ho)
    _check_index(ops, patch, Ho, Wo, stride)
    fast = ops.mask_to_index(patch.to(DEV), Ho, Wo, stride)
    monkeypatch.setenv("LDN_INDEX_GENERIC", "1")
    gen = ops.mask_to_index(patch.to(DEV), Ho, Wo, stride)
    n3, n1 = int(gen.cnt[0]), int(gen.cnt[1])
    assert torch.equal(fast.cnt, gen.cnt) and torch.equal(fast.pre3, gen.pre3) and torch.equal(fast.pre1, gen.pre1)
    assert torch.equal(fast.stats, gen.stats)
    assert torch.equal(fast.pos3, gen.pos3) and torch.equal(fast.pos1, gen.pos1)
    assert torch.equal(fast.idx3[:n3], gen.idx3[:n3]) and torch.equal(fast.idx1[:n1], gen.idx1[:n1])
    assert torch.equal(fast.nbr[:n3 * 9], gen.nbr[:n3 * 9])


def test_gather_scatter(ops):
    rows_total, C = 500, 64
    src = seeded_randn((rows_total, C), 3).to(DEV)
    rows = torch.randperm(rows_total, generator=torch.Generator().manual_seed(1))[:123].sort().values.to(torch.int32)
    count = torch.tensor([100], dtype=torch.int32, device=DEV)
    packed = ops.gather_rows(src, rows.to(DEV), count=count)
    assert torch.equal(packed[:100].cpu(), src.cpu()[rows[:100].long()])
    ident = seeded_randn((rows_total, C), 4).to(DEV)
    out = ident.clone()
    ops.scatter_add_relu(packed, rows.to(DEV), ident, out, count=count)
    want = ident.cpu().clone()
    want[rows[:100].long()] = torch.relu(ident.cpu()[rows[:100].long()] + packed[:100].cpu())
    assert torch.equal(out.cpu(), want)


# ------------------------------------------------------------------ maskers
@pytest.mark.parametrize("cin,g,S,hin", [(16, 1, 4, 8), (16, 1, 8, 8), (16, 2, 4, 8), (8, 1, 3, 14), (8, 1, 1, 7),
                                         (256, 1, 7, 56), (64, 1, 14, 56), (64, 1, 1, 14), (256, 2, 1, 28)])
def test_spatial_masker(ops, cin, g, S, hin):
    torch.manual_seed(cin + S)
    ref = TR.SpatialMaskerRef(cin, g, S).eval()
    with torch.no_grad():
        ref.conv.weight.normal_()
        ref.conv.bias.normal_()
    x = F.relu(seeded_randn((4, cin, hin, hin), 5))
    with torch.no_grad():
        want_logits = ref.logits(x).reshape(4, 2 * g, *ref.logits(x).shape[-2:])
        want_mask = (want_logits[:, :g] >= want_logits[:, g:]).float()
    xn = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    mask, logits = ops.spatial_masker(xn, ref.conv.weight.detach().reshape(2 * g, cin).to(DEV).contiguous(),
                                      ref.conv.bias.detach().to(DEV), g, S, want_logits=True)
    assert torch.allclose(logits.cpu(), want_logits, atol=1e-4, rtol=1e-5)  # fp32 reduction-order tolerance
    margin = (want_logits[:, :g] - want_logits[:, g:]).abs()
    differs = mask.cpu() != want_mask
    assert not bool((differs & (margin > 1e-4)).any()), "mask decisions may differ only at near-ties"


@pytest.mark.parametrize("cin,g,hin,win", [(256, 1, 14, 14), (256, 2, 9, 20), (512, 1, 7, 7), (1024, 1, 5, 3), (2048, 2, 7, 7), (64, 1, 12, 12), (16, 2, 6, 5), (320, 1, 7, 7)])
def test_pixel_masker_equals_general_kernel(ops, cin, g, hin, win):
    """Round 6: per-pixel masks (mask_size == the map, BASELINE config 1) run on k_pixel_masker (eight pixels' rows in flight per wave).  Same
    arithmetic per pixel as k_spatial_masker: logits and decisions bit-equal (the general kernel is pinned to the oracle above), ragged pixel
    counts and non-square maps included."""
    import os
    torch.manual_seed(cin + hin)
    w = torch.randn(2 * g, cin, device=DEV)
    b = torch.randn(2 * g, device=DEV)
    xn = F.relu(seeded_randn((3, hin, win, cin), 11)).to(DEV).contiguous()
    mask, logits = ops.spatial_masker(xn, w, b, g, hin, want_logits=True)
    os.environ["LDN_PIXEL_MASKER_OLD"] = "1"
    try:
        mask0, logits0 = ops.spatial_masker(xn, w, b, g, hin, want_logits=True)
    finally:
        del os.environ["LDN_PIXEL_MASKER_OLD"]
    assert torch.equal(logits, logits0) and torch.equal(mask, mask0)
    want = torch.einsum("bhwc,oc->bohw", xn.double(), w.double()) + b.double().view(1, -1, 1, 1)
    assert torch.allclose(logits.double(), want, atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("hin,S", [(14, 3), (52, 7), (14, 7), (28, 7)])
def test_spatial_masker_patch_carry_uneven_grid(ops, hin, S):
    """on an uneven grid (H % S != 0) the adaptive-pool bin of a patch overlaps pixels the nearest mapping gives to a
    NEIGHBOURING patch.  A previous block that rewrote only the neighbour leaves the patch's own carry flag at 0 although its bin changed:
    the carry must not be taken there (ops gates it, the library refuses it); on even grids it stays bit-identical to a fresh pass."""
    import laudnet_amd._lib as L
    cin, B = 16, 3
    w = seeded_randn((2, cin), 3).to(DEV)
    b = torch.zeros(2, device=DEV)
    x0 = F.relu(seeded_randn((B, hin, hin, cin), 11)).to(DEV).contiguous()
    _, lg0, work = ops.spatial_masker(x0, w, b, 1, S, want_logits=True, return_work=True)
    # the "previous block" executed a checkerboard patch mask: rewrite exactly the pixels its conv3 would (nearest mapping)
    pm = ((torch.arange(S)[:, None] + torch.arange(S)[None, :]) % 2).float().expand(B, S, S).contiguous()
    iy = (torch.arange(hin) * S) // hin
    pix = pm[:, iy][:, :, iy].to(DEV)                              # [B, H, W] nearest up-sampling (laud_resnet.py:106)
    x1 = (x0 + 3.0 * pix[..., None]).contiguous()
    _, want = ops.spatial_masker(x1, w, b, 1, S, want_logits=True)
    _, got = ops.spatial_masker(x1, w, b, 1, S, want_logits=True, carry=(work, None, work.ldn_shape_key, pm.to(DEV)))
    assert torch.equal(got, want), (hin, S)
    assert not torch.equal(want, lg0)
    if hin % S:
        lib = L.load()
        mask = torch.empty(B, 1, S, S, device=DEV)
        rc = lib.ldn_spatial_masker(L.ptr(x1), B, hin, hin, cin, L.ptr(w), L.ptr(b), 1, S, L.ptr(mask), None, L.ptr(work), None,
                                    L.ptr(pm.to(DEV)), L.stream_ptr())
        assert rc != 0 and b"patch carry" in lib.ldn_last_error()
    # a carry whose shape key was lost (work tensor re-created by .to() / .view) is dropped, not a TypeError
    _, got2 = ops.spatial_masker(x1, w, b, 1, S, want_logits=True, carry=(work, None, None, pm.to(DEV)))
    assert torch.equal(got2, want)


def test_spatial_masker_tie_keeps(ops):
    x = torch.rand(1, 2, 2, 4, device=DEV)
    mask, _ = ops.spatial_masker(x, torch.zeros(2, 4, device=DEV), torch.zeros(2, device=DEV), 1, 2)
    assert float(mask.min()) == 1.0  # models/utils.py:60: >= keeps ties


@pytest.mark.parametrize("cin,G,layers,gran,hw", [(32, 8, 2, 1, 6), (32, 8, 1, 2, 6), (256, 32, 2, 2, 56),
                                                  (1024, 128, 2, 2, 14), (2048, 256, 2, 2, 7), (64, 16, 2, 4, 9)])
def test_channel_masker(ops, cin, G, layers, gran, hw):
    torch.manual_seed(G + layers)
    ref = TR.ChannelMaskerMLPRef(cin, G, layers=layers).eval()
    with torch.no_grad():
        for prm in ref.parameters():
            prm.normal_()
    B = 5
    x = F.relu(seeded_randn((B, cin, hw, hw), 6))
    with torch.no_grad():
        want_logits = ref.logits(x).reshape(B, 2 * G)
    if layers == 2:
        w1, b1, w2, b2 = ref.conv[0].weight, ref.conv[0].bias, ref.conv[2].weight, ref.conv[2].bias
    else:
        w1, b1, w2, b2 = ref.conv.weight, ref.conv.bias, None, None
    d = lambda t: None if t is None else t.detach().to(DEV).contiguous()
    xn = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    mask, idx, cnt, logits = ops.channel_masker(xn, d(w1), d(b1), d(w2), d(b2), G, gran, want_logits=True)
    scale = want_logits.abs().max().item()
    assert torch.allclose(logits.cpu(), want_logits, atol=2e-5 * max(scale, 1.0), rtol=1e-5)
    want_mask = (want_logits[:, :G] >= want_logits[:, G:]).float()
    margin = (want_logits[:, :G] - want_logits[:, G:]).abs()
    assert not bool(((mask.cpu() != want_mask) & (margin > 1e-4 * max(scale, 1.0))).any())
    # the lists must be exactly the lists of the mask the kernel decided on
    widx, wcnt = IR.channel_lists(mask.cpu().numpy(), G * gran)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)
    got = idx.cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b, :wcnt[b]], widx[b, :wcnt[b]])
    # injected-mask variant: lists only
    m_in = seeded_bernoulli((B, G), 0.6, 9)
    mask2, idx2, cnt2, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=m_in.to(DEV))
    widx, wcnt = IR.channel_lists(m_in.numpy(), G * gran)
    assert torch.equal(mask2.cpu(), m_in) and np.array_equal(cnt2.cpu().numpy(), wcnt)
    for b in range(B):
        assert np.array_equal(idx2.cpu().numpy()[b, :wcnt[b]], widx[b, :wcnt[b]])


# ------------------------------------------------------------------ packed-row convolution
def _affine(c, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1


@pytest.mark.parametrize("rows,cin,cout", [(300, 64, 64), (1000, 256, 64), (257, 64, 256), (129, 8, 16), (64, 36, 200),
                                           (5000, 128, 128),
                                           # widths that are not multiples of 32 (LAD-RegNet 144 / 784): zero-filled K tail, ragged column tile
                                           (700, 144, 144), (300, 784, 144), (513, 144, 784), (300, 24, 40), (256, 72, 36),
                                           (400, 320, 784), (290, 48, 168),
                                           # 160-column tiles (k_dense<5>): whole tiles, the pinned schedule, and a single tile
                                           (600, 320, 320), (300, 64, 160), (1000, 160, 480)])
def test_conv_rows_1x1_gather(ops, rows, cin, cout):
    a = seeded_randn((rows, cin), 1)
    w = seeded_randn((cout, 1, cin), 2) * (2.0 / cin) ** 0.5
    sc, sh = _affine(cout, 3)
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(4))[: rows * 2 // 3].sort().values
    count = torch.tensor([len(perm) - 5], dtype=torch.int32)
    out = torch.full((rows, cout), -7.0, device=DEV)
    ops.conv_rows(a.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV), out, a_rows=perm.to(torch.int32).to(DEV), taps=1,
                  m_count=count.to(DEV), m_cap=rows, relu=1)
    n = int(count)
    want = torch.relu((a[perm[:n]].double() @ w[:, 0].double().T) * sc.double() + sh.double()).float()
    assert torch.allclose(out[:n].cpu(), want, atol=1e-4, rtol=1e-4)   # fp32 MFMA vs fp64 reference
    assert bool((out[n:] == -7.0).all()), "rows beyond the device-side count must not be written"


@pytest.mark.parametrize("B,H,C,cout,stride", [(2, 14, 16, 16, 1), (3, 14, 64, 64, 2), (2, 7, 128, 128, 1), (2, 14, 48, 72, 1), (2, 14, 144, 40, 2)])
def test_conv_rows_3x3_table_scatter(ops, B, H, C, cout, stride):
    """Full spatial-mode slice at op level: mask -> index -> 3x3 through the neighbour table, and the final
    1x1 with residual scatter-add, against F.conv2d on masked dense tensors."""
    Ho = H // stride if stride > 1 else H
    Hi = Ho * stride
    patch = seeded_bernoulli((B, Ho, Ho), 0.5, 11)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    h1_dense = seeded_randn((B, C, Hi, Hi), 12)
    m1 = torch.from_numpy(IR.dilate_mask(patch.numpy() > 0.5, stride, 1)).float().unsqueeze(1)
    idx1 = ix.idx1[: int(ix.cnt[1])].long().cpu()
    h1_rows = h1_dense.permute(0, 2, 3, 1).reshape(-1, C)
    packed_h1 = torch.zeros(ix.cap1, C)
    packed_h1[: len(idx1)] = h1_rows[idx1]
    w = seeded_randn((cout, C, 3, 3), 13) * (2.0 / (9 * C)) ** 0.5
    sc, sh = _affine(cout, 14)
    out = torch.zeros(ix.cap3, cout, device=DEV)
    ops.conv_rows(packed_h1.to(DEV), w.permute(0, 2, 3, 1).reshape(cout, 9, C).contiguous().to(DEV), sc.to(DEV),
                  sh.to(DEV), out, a_rows=ix.nbr, taps=9, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1)
    dense = F.conv2d((h1_dense * m1).double(), w.double(), stride=stride, padding=1)
    dense = torch.relu(dense * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float()
    idx3 = ix.idx3[: int(ix.cnt[0])].long().cpu()
    want = dense.permute(0, 2, 3, 1).reshape(-1, cout)[idx3]
    assert torch.allclose(out[: len(idx3)].cpu(), want, atol=1e-4, rtol=1e-4)
    # scatter-add epilogue
    ident = seeded_randn((B * Ho * Ho, cout), 15)
    res = ident.clone().to(DEV)
    w3 = seeded_randn((cout, 1, cout), 16) * (2.0 / cout) ** 0.5
    ops.conv_rows(out, w3.to(DEV), sc.to(DEV), sh.to(DEV), res, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1,
                  out_rows=ix.idx3, residual2d=res)
    want_full = ident.clone()
    y = (want.double() @ w3[:, 0].double().T) * sc.double() + sh.double()
    want_full[idx3] = torch.relu(ident[idx3].double() + y).float()
    assert torch.allclose(res.cpu(), want_full, atol=2e-4, rtol=1e-4)
    # scale == NULL: weights carry the BN scale, the residual tile starts the accumulators (in-place scatter form)
    res2 = ident.clone().to(DEV)
    ops.conv_rows(out, (w3 * sc.view(-1, 1, 1)).to(DEV), None, sh.to(DEV), res2, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3,
                  relu=1, out_rows=ix.idx3, residual2d=res2)
    assert torch.allclose(res2.cpu(), want_full, atol=2e-4, rtol=1e-4)


# ------------------------------------------------------------------ per-image convolution
@pytest.mark.parametrize("B,H,cin,cout,ksize,stride", [
    (2, 14, 64, 64, 1, 1), (3, 14, 32, 48, 3, 1), (2, 28, 64, 128, 1, 2), (2, 14, 16, 16, 3, 2), (9, 7, 128, 256, 3, 1),
    (2, 56, 64, 64, 3, 1), (2, 13, 8, 24, 3, 2),
])
def test_conv_image_dense(ops, B, H, cin, cout, ksize, stride):
    x = seeded_randn((B, cin, H, H), 21)
    w = seeded_randn((cout, cin, ksize, ksize), 22) * (2.0 / (cin * ksize * ksize)) ** 0.5
    sc, sh = _affine(cout, 23)
    Ho = (H - 1) // stride + 1
    out = torch.empty(B, Ho, Ho, cout, device=DEV)
    resid = seeded_randn((B, Ho, Ho, cout), 24)
    colsum = torch.full((B, (Ho * Ho + 31) // 32, cout), float("nan"), device=DEV)
    ops.conv_image(x.permute(0, 2, 3, 1).contiguous().to(DEV),
                   w.permute(0, 2, 3, 1).reshape(cout, ksize * ksize, cin).contiguous().to(DEV), sc.to(DEV), sh.to(DEV),
                   out, ksize=ksize, stride=stride, relu=1, residual=resid.to(DEV), colsum=colsum)
    want = F.conv2d(x.double(), w.double(), stride=stride, padding=ksize // 2)
    want = want * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    want = torch.relu(want.permute(0, 2, 3, 1) + resid.double()).float()
    assert torch.allclose(out.cpu(), want, atol=1e-4, rtol=1e-4)
    # scale == NULL: BN scale multiplied into the weights by the caller, residual-initialised accumulators
    out2 = torch.full((B, Ho, Ho, cout), float("nan"), device=DEV)
    ws = w * sc.view(-1, 1, 1, 1)
    ops.conv_image(x.permute(0, 2, 3, 1).contiguous().to(DEV),
                   ws.permute(0, 2, 3, 1).reshape(cout, ksize * ksize, cin).contiguous().to(DEV), None, sh.to(DEV),
                   out2, ksize=ksize, stride=stride, relu=1, residual=resid.to(DEV))
    assert torch.allclose(out2.cpu(), want, atol=1e-4, rtol=1e-4)
    # fused GAP partials: sums over each run of 32 pixels (row-major), every slot written exactly once
    flat = out.cpu().reshape(B, Ho * Ho, cout).double()
    pad = (-flat.shape[1]) % 32
    flat = torch.cat([flat, flat.new_zeros(B, pad, cout)], dim=1).reshape(B, -1, 32, cout).sum(dim=2)
    assert torch.allclose(colsum.cpu().double(), flat, atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("B,H,cin,W,gran,stride", [(3, 14, 64, 16, 1, 1), (3, 14, 32, 16, 2, 2), (4, 7, 128, 64, 4, 1),
                                                    (2, 28, 64, 32, 2, 1), (5, 14, 256, 256, 2, 1),
                                                    # the tile shapes of the R101 stages: 6x2 / 4x4 / 8x6 / 2x10 subtiles
                                                    (2, 56, 64, 64, 2, 1), (3, 28, 128, 128, 2, 1), (3, 7, 512, 512, 2, 1),
                                                    (3, 14, 256, 512, 2, 2)])
def test_conv_image_channel_subsets(ops, B, H, cin, W, gran, stride):
    """conv1 (output subset) -> conv2 (input+output subsets, border-class shift table) -> conv3 (input subset),
    against the dense-emulation algebra of laud_resnet.py:115-133 (mask before BN)."""
    G = W // gran
    gm = seeded_bernoulli((B, G), 0.6, 31)
    gm[0] = 0.0          # an image with no active channel
    gm[1] = 1.0          # an image with all channels
    blk = TR.BottleneckRef(cin, W, stride=stride, downsample=None, dyn_mode="channel", channel_dyn_granularity=gran,
                           channel_masker="MLP", output_size=H // stride).eval()
    TR.randomize_bn_(blk, 5)
    with torch.no_grad():
        for m in (blk.conv1, blk.conv2, blk.conv3):
            m.weight.normal_(0, (2.0 / (m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3])) ** 0.5)
    x = F.relu(seeded_randn((B, cin, H, H), 32))
    cm = TR.broadcast_channel_mask(gm, W)
    with torch.no_grad():
        h1 = F.relu(blk.bn1(blk.conv1(x) * cm))
        h2 = F.relu(blk.bn2(blk.conv2(h1) * cm))
        y3 = blk.bn3(blk.conv3(h2))
    # HIP path with the folded tables of laudnet_amd.laud_resnet.Bottleneck._prepare
    from laudnet_amd.laud_resnet import Bottleneck
    hb = Bottleneck(cin, W, stride=stride, downsample=None, dyn_mode="channel", channel_dyn_granularity=gran,
                    channel_masker="MLP", output_size=H // stride).eval()
    hb.load_state_dict(blk.state_dict())
    hb = hb.to(DEV)
    p = hb._prepare(torch.device(DEV))
    _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm.to(DEV))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    Ho = (H - 1) // stride + 1
    g1 = torch.full((B, H, H, W), float("nan"), device=DEV)
    ops.conv_image(xn, p["w1"], p["s1"], p["t1"], g1, n_idx=idx, n_cnt=cnt, post_sub=p["c1"], relu=1)
    g2 = torch.full((B, Ho, Ho, W), float("nan"), device=DEV)
    ops.conv_image(g1, p["w2"], p["s2"], p["t2_tab"], g2, ksize=3, stride=stride, k_idx=idx, k_cnt=cnt, kgran=gran,
                   n_idx=idx, n_cnt=cnt, post_sub=p["c2"], relu=1)
    g3 = torch.empty(B, Ho, Ho, 4 * W, device=DEV)  # p['w2'] / p['w3'] are k-major here (channel mode)
    ops.conv_image(g2, p["w3"], None, p["t3c"], g3, k_idx=idx, k_cnt=cnt, kgran=gran, relu=0)
    # unpack the left-packed h1/h2 and compare on the active channels
    cidx, ccnt = idx.cpu().long(), cnt.cpu()
    c1, c2 = p["c1"].cpu(), p["c2"].cpu()
    for b in range(B):
        n = int(ccnt[b])
        ch = cidx[b, :n]
        want1 = h1[b, ch].permute(1, 2, 0) - c1[ch]
        assert torch.allclose(g1[b, :, :, :n].cpu(), want1, atol=1e-4, rtol=1e-4), f"conv1 image {b}"
        want2 = h2[b, ch].permute(1, 2, 0) - c2[ch]
        assert torch.allclose(g2[b, :, :, :n].cpu(), want2, atol=2e-4, rtol=1e-4), f"conv2 image {b}"
        pad_hi = min((n + 3) // 4 * 4, W)
        assert bool((g1[b, :, :, n:pad_hi] == 0).all()) and bool((g2[b, :, :, n:pad_hi] == 0).all())
    assert torch.allclose(g3.cpu(), y3.permute(0, 2, 3, 1), atol=5e-4, rtol=1e-4)


# ------------------------------------------------------------------ the wide 1x1 convs (k_conv1x1_stream in bf16x3 mode)
def _ref_rows(a, w, sc, sh, a_rows=None):
    x = a if a_rows is None else a[a_rows.long()]
    y = x.double() @ w[:, 0].double().T
    if sc is not None:
        y = y * sc.double()
    return y + sh.double()


@pytest.mark.parametrize("rows,cin,cout,count", [(1500, 64, 256, 1500), (12544, 512, 2048, 12544), (3000, 96, 384, 1777),
                                                 (600, 160, 512, 0), (200, 1024, 256, 130)])
def test_wide_1x1_rows_scatter_residual(ops, rows, cin, cout, count):
    """conv3 of the spatial / layer path at the widths that take the persistent streaming kernel in bf16x3 mode
    (cout % 128 == 0, cout >= 256): gathered A rows, device-side row count, scatter-add into the residual stream in place,
    BN scale folded into the weights (scale == NULL) and as a separate vector; rows beyond the count untouched.
    The 12544-row case is the stage-4 dense execution of the headline model (N tiles split over workgroups)."""
    src_rows = rows + 37
    a = seeded_randn((src_rows, cin), 51)
    w = seeded_randn((cout, 1, cin), 52) * (2.0 / cin) ** 0.5
    sc, sh = _affine(cout, 53)
    g = torch.Generator().manual_seed(54)
    a_rows = torch.randperm(src_rows, generator=g)[:rows].to(torch.int32)
    out_rows = torch.randperm(rows, generator=g).to(torch.int32)
    ident = seeded_randn((rows, cout), 55)
    cnt = torch.tensor([count], dtype=torch.int32, device=DEV)
    want = ident.clone()
    y = _ref_rows(a, w, sc, sh, a_rows[:count])
    want[out_rows[:count].long()] = torch.relu(ident[out_rows[:count].long()].double() + y).float()
    for prescaled in (True, False):
        res = ident.clone().to(DEV)
        if prescaled:
            ops.conv_rows(a.to(DEV), (w * sc.view(-1, 1, 1)).to(DEV), None, sh.to(DEV), res, a_rows=a_rows.to(DEV), taps=1,
                          m_count=cnt, m_cap=rows, relu=1, out_rows=out_rows.to(DEV), residual2d=res)
        else:   # separate scale + residual: stays on the general kernel (same contract)
            ops.conv_rows(a.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV), res, a_rows=a_rows.to(DEV), taps=1,
                          m_count=cnt, m_cap=rows, relu=1, out_rows=out_rows.to(DEV), residual2d=res)
        assert torch.allclose(res.cpu(), want, atol=2e-4, rtol=1e-4), f"prescaled={prescaled}"


@pytest.mark.parametrize("rows,cin,cout", [(1000, 64, 256), (5000, 256, 512)])
def test_wide_1x1_rows_conditional_relu(ops, rows, cin, cout):
    """The strided projection shortcut of the spatial path: separate scale vector, no residual, ReLU only on the rows whose
    flag is negative (relu == 2)."""
    a = seeded_randn((rows, cin), 61)
    w = seeded_randn((cout, 1, cin), 62) * (2.0 / cin) ** 0.5
    sc, sh = _affine(cout, 63)
    flag = torch.where(seeded_bernoulli((rows,), 0.5, 64) > 0.5, torch.tensor(-1), torch.tensor(3)).to(torch.int32)
    out = torch.full((rows, cout), float("nan"), device=DEV)
    ops.conv_rows(a.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV), out, taps=1, m_cap=rows, relu=2, relu_if_neg=flag.to(DEV))
    y = _ref_rows(a, w, sc, sh)
    want = torch.where((flag < 0).view(-1, 1), torch.relu(y), y).float()
    assert torch.allclose(out.cpu(), want, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("B,H,W,cout,gran,p", [(4, 14, 256, 1024, 2, 0.62), (3, 28, 128, 512, 2, 0.5), (2, 56, 64, 256, 1, 0.6),
                                               (5, 14, 64, 256, 2, 0.0), (3, 10, 32, 384, 4, 1.0)])
def test_wide_1x1_image_gathered_inputs(ops, B, H, W, cout, gran, p):
    """conv3 of channel mode: left-packed input columns (garbage beyond roundup4(count)), k-major weights gathered by the
    per-image channel list (images with NO active channel included), residual in place, fused GAP partials."""
    G = W // gran
    gm = seeded_bernoulli((B, G), p, 71)
    _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm.to(DEV))
    h2 = seeded_randn((B, H, H, W), 72)
    cidx, ccnt = idx.cpu().long(), cnt.cpu()
    for b in range(B):
        n = int(ccnt[b])
        h2[b, :, :, n:] = float("nan")
        h2[b, :, :, n:min((n + 3) // 4 * 4, W)] = 0
    w = seeded_randn((1, W, cout), 73) * (2.0 / W) ** 0.5
    _, sh = _affine(cout, 74)
    x = seeded_randn((B, H, H, cout), 75)
    out = x.clone().to(DEV)
    colsum = torch.full((B, (H * H + 31) // 32, cout), float("nan"), device=DEV)
    ops.conv_image(h2.to(DEV), w.to(DEV), None, sh.to(DEV), out, k_idx=idx, k_cnt=cnt, kgran=gran, relu=1, residual=out,
                   colsum=colsum)
    want = torch.empty(B, H, H, cout, dtype=torch.float64)
    for b in range(B):
        n = int(ccnt[b])
        ch = cidx[b, :n]
        want[b] = torch.relu(h2[b, :, :, :n].double() @ w[0, ch].double() + sh.double() + x[b].double())
    assert torch.allclose(out.cpu(), want.float(), atol=2e-4, rtol=1e-4)
    flat = out.cpu().reshape(B, H * H, cout).double()
    pad = (-flat.shape[1]) % 32
    flat = torch.cat([flat, flat.new_zeros(B, pad, cout)], dim=1).reshape(B, -1, 32, cout).sum(dim=2)
    assert torch.allclose(colsum.cpu().double(), flat, atol=1e-3, rtol=1e-5)


# ------------------------------------------------------------------ ExpandMask module surface (models/utils.py:67-89)
def test_expand_mask_module_vs_reference_fixtures(ops):
    """The HIP-side ExpandMask (dilation folded into ldn_mask_to_index; with several mask groups every output group is the OR
    of all input groups, utils.py:81) against the outputs of the reference's ExpandMask stored in l1_ops.pt."""
    from helpers import load_golden
    from laudnet_amd.laud_resnet import ExpandMask
    for case in load_golden("l1_ops.pt")["expand"]:
        m = case["mask"]
        if case["padding"] == 0 and case["stride"] != 1:
            continue    # pure zero-insertion: never instantiated by a block (expander2 is (1,0), expander1 is (stride,1))
        got = ExpandMask(stride=case["stride"], padding=case["padding"], mask_channel_group=m.shape[1])(m.to(DEV))
        assert torch.equal(got.cpu(), case["y"].bool()), (case["stride"], case["padding"], tuple(m.shape))


@pytest.mark.parametrize("B,H,C,cout,stride", [(2, 14, 32, 64, 1), (3, 14, 64, 64, 2)])
def test_dense_kernel_3x3_neighbour_table(ops, B, H, C, cout, stride, math_mode):
    """k_dense's 3x3 form (ldn_conv_rows_split / ldn_conv_rows_f32, taps == 9): same result as round 1's kernel on the spatial-mode slice
    mask -> index -> 3x3 through the neighbour table, in both arithmetic modes (round 4: the fp32 mode has its own k_dense form, opt-in)."""
    f32_before, taps_before = ops.USE_DENSE_F32, ops.DENSE_TAPS
    ops.USE_DENSE_F32 = True
    Ho = H // stride if stride > 1 else H
    patch = seeded_bernoulli((B, Ho, Ho), 0.5, 11)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    h1 = seeded_randn((ix.cap1, C), 12).to(DEV)
    w = (seeded_randn((cout, 9, C), 13) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    sc, sh = _affine(cout, 14)
    outs = []
    for taps_set in ((1,), (1, 9)):
        ops.DENSE_TAPS = taps_set
        try:
            out = torch.zeros(ix.cap3, cout, device=DEV)
            ops.conv_rows(h1, w, sc.to(DEV), sh.to(DEV), out, a_rows=ix.nbr, taps=9, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1)
            outs.append(out.cpu())
        finally:
            ops.DENSE_TAPS = taps_before          # (restored, not reset: later tests compare against the k_dense 3x3 form bit for bit)
            ops.USE_DENSE_F32 = f32_before if taps_set == (1, 9) else True
    n = int(ix.cnt[0])
    assert torch.allclose(outs[0][:n], outs[1][:n], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,Ho,stride,C,p", [(3, 14, 1, 144, 0.5), (2, 14, 2, 64, 0.6), (2, 7, 1, 320, 1.0), (4, 9, 2, 784, 0.4), (1, 5, 1, 16, 1.0)])
def test_grouped16_conv3x3_rows_mfma_vs_torch(ops, B, Ho, stride, C, p):
    """Matrix-core grouped 3x3 (group width 16, bf16x3) over packed rows through the neighbour table, against F.conv2d(groups) on the
    dense map evaluated at the active pixels; inactive input pixels are absent from the packed input exactly as in the block."""
    import torch.nn.functional as F
    Hi = Ho * stride
    patch = seeded_bernoulli((B, Ho, Ho), p, 91 + C)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    n3, n1 = int(ix.cnt[0].item()), int(ix.cnt[1].item())
    x = seeded_randn((B, C, Hi, Hi), 92 + C)
    w = seeded_randn((C, 16, 3, 3), 93) * 0.1
    scale = 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(94))
    shift = 0.1 * seeded_randn((C,), 95)
    m1 = torch.zeros(B * Hi * Hi, dtype=torch.bool)
    m1[ix.idx1[:n1].long().cpu()] = True
    xm = x * m1.view(B, 1, Hi, Hi)                     # the packed input holds the dilated-mask pixels only; the others read as zero
    want = F.relu(F.conv2d(xm, w, None, stride, 1, 1, C // 16) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    want = want.permute(0, 2, 3, 1).reshape(B * Ho * Ho, C)[ix.idx3[:n3].long().cpu()]
    a = x.permute(0, 2, 3, 1).reshape(B * Hi * Hi, C)[ix.idx1[:n1].long().cpu()].contiguous().to(DEV)
    a = torch.cat([a, torch.zeros(max(ix.cap1 - n1, 0), C, device=DEV)])
    frag = ops.pack_grouped16_weights(w.permute(0, 2, 3, 1).reshape(C, 9, 16).contiguous().to(DEV))
    out = torch.full((ix.cap3, C), float("nan"), device=DEV)
    ops.grouped16_conv3x3_rows(a, ix.nbr, frag, scale.to(DEV), shift.to(DEV), out, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1)
    got = out[:n3].cpu()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("B,Ho,stride,C,p", [(6, 14, 1, 320, 0.5), (3, 14, 2, 64, 0.7), (4, 7, 1, 784, 0.5), (5, 14, 2, 320, 0.4),
                                             (3, 28, 1, 144, 0.6), (2, 5, 1, 16, 1.0), (4, 9, 2, 48, 0.0),
                                             # maps beyond one workgroup's LDS: bands of output rows (halo rows re-staged)
                                             (3, 56, 2, 64, 0.7), (3, 28, 2, 144, 0.7), (2, 56, 1, 32, 1.0), (2, 37, 2, 16, 1.0)])
def test_grouped16_conv3x3_whole_images_vs_rows_kernel(ops, B, Ho, stride, C, p):
    """Layer skip (one decision per image): the LDS-staged whole-image form of the grouped 3x3 (ldn_grouped16_conv3x3_images)
    against the neighbour-table kernel on the same packed rows (same products, same order over taps: identical up to the fp32
    rounding of the two bf16x3 sums -- they are bit-equal) and against F.conv2d(groups) on the kept images."""
    import torch.nn.functional as F
    Hi = Ho * stride
    patch = seeded_bernoulli((B, 1, 1), p, 31 + C + B)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    n3, n1 = int(ix.cnt[0].item()), int(ix.cnt[1].item())
    assert ops.grouped16_images_fit(Hi, Hi, C) > 0
    x = seeded_randn((B, C, Hi, Hi), 32 + C)
    w = seeded_randn((C, 16, 3, 3), 33 + C) * (2.0 / 144) ** 0.5
    sc, sh = _affine(C, 34)
    rows_in = x.permute(0, 2, 3, 1).reshape(-1, C)
    h_a = torch.full((ix.cap1, C), float("nan"))          # rows beyond the kept images must never be read
    if n1:
        h_a[:n1] = rows_in[ix.idx1[:n1].long().cpu()]
    frag = ops.pack_grouped16_weights(w.permute(0, 2, 3, 1).reshape(C, 9, 16).contiguous().to(DEV))
    out_img = torch.full((ix.cap3, C), -7.0, device=DEV)
    ops.grouped16_conv3x3_images(h_a.to(DEV), frag, sc.to(DEV), sh.to(DEV), out_img, m_count=ix.cnt[0:1],
                                 images=(B, Hi, Hi, Ho, Ho, stride), relu=1)
    out_rows = torch.full((ix.cap3, C), -7.0, device=DEV)
    ops.grouped16_conv3x3_rows(torch.nan_to_num(h_a).to(DEV), ix.nbr, frag, sc.to(DEV), sh.to(DEV), out_rows, m_count=ix.cnt[0:1],
                               m_cap=ix.cap3, relu=1)
    assert torch.equal(out_img[:n3], out_rows[:n3])
    assert bool((out_img[n3:] == -7.0).all()), "rows beyond the kept images must not be written"
    dense = F.conv2d(x.double(), w.double(), stride=stride, padding=1, groups=C // 16)
    dense = torch.relu(dense * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float()
    want = dense.permute(0, 2, 3, 1).reshape(-1, C)[ix.idx3[:n3].long().cpu()]
    assert torch.allclose(out_img[:n3].cpu(), want, atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------ k_dense<8>: 256 x 256 tiles at full-size row counts
@pytest.mark.parametrize("rows,cin,cout,count", [(50176, 256, 512, None), (100001, 64, 256, 99001), (100352, 40, 256, None)])
def test_dense_256_tiles_fullsize_rows_vs_fp64(ops, rows, cin, cout, count):
    """Shared-weight 1x1 rows on k_dense's 256-column tiles at the row counts of the bench (>= 384 tiles): against fp64; ragged last M
    tile, device-side count, K not a multiple of 32, row gather + scatter + conditional ReLU + scale vector."""
    ops.set_math_mode("bf16x3")
    try:
        a = seeded_randn((rows, cin), 1)
        w = seeded_randn((cout, 1, cin), 2) * (2.0 / cin) ** 0.5
        sh = seeded_randn((cout,), 3)
        res = seeded_randn((rows, cout), 4)
        out = torch.zeros(rows, cout, device=DEV)
        ops.conv_rows(a.to(DEV), w.to(DEV), None, sh.to(DEV), out, taps=1, m_cap=rows, relu=1, residual2d=res.to(DEV))
        want = torch.relu(a.double() @ w[:, 0].double().T + sh.double() + res.double()).float()
        assert torch.allclose(out.cpu(), want, atol=1e-4, rtol=1e-4)
        if count is not None:
            g = torch.Generator().manual_seed(7)
            src = torch.randint(0, rows, (rows,), generator=g).to(torch.int32)
            dst = torch.randperm(rows, generator=g).to(torch.int32)
            rneg = torch.where(torch.rand(rows, generator=g) < 0.5, -1, 1).to(torch.int32)
            sc = seeded_randn((cout,), 5).abs() + 0.5
            out2 = torch.full((rows, cout), -7.0, device=DEV)
            cnt = torch.tensor([count], dtype=torch.int32, device=DEV)
            ops.conv_rows(a.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV), out2, a_rows=src.to(DEV), taps=1, m_count=cnt, m_cap=rows, relu=2,
                          relu_if_neg=rneg.to(DEV), out_rows=dst.to(DEV))
            y = (a[src[:count].long()].double() @ w[:, 0].double().T) * sc.double() + sh.double()
            y = torch.where((rneg[:count] < 0).view(-1, 1), torch.relu(y), y).float()
            want2 = torch.full((rows, cout), -7.0)
            want2[dst[:count].long()] = y
            assert torch.allclose(out2.cpu(), want2, atol=1e-4, rtol=1e-4)
    finally:
        ops.set_math_mode(None)


@pytest.mark.parametrize("rows,cin,cout,taps", [(1000, 64, 256, 1), (700, 256, 64, 1), (513, 40, 144, 1), (600, 32, 128, 9)])
def test_conv_rows_f32_kernel_vs_fp64(ops, rows, cin, cout, taps):
    """ldn_conv_rows_f32 (k_dense in true-fp32 MFMA arithmetic, opt-in: ops.USE_DENSE_F32): gathered rows, ragged widths, the 3x3
    neighbour-table form, residual + ReLU -- against fp64 at fp32 round-off."""
    from laudnet_amd import ops as _o
    before = (_o.USE_DENSE_F32, _o.get_math_mode())
    _o.USE_DENSE_F32 = True
    _o.set_math_mode("fp32")
    try:
        a = seeded_randn((rows, cin), 3)
        w = seeded_randn((cout, taps, cin), 4) * (1.0 / (taps * cin)) ** 0.5
        sh = seeded_randn((cout,), 5) * 0.1
        res = seeded_randn((rows, cout), 6)
        if taps == 1:
            src = torch.randperm(rows, generator=torch.Generator().manual_seed(7)).to(torch.int32)
            want = torch.relu(a.double()[src.long()] @ w.double().reshape(cout, cin).t() + sh.double() + res.double())
            a_rows = src.to(DEV)
        else:
            nb = torch.randint(-1, rows, (rows, 9), generator=torch.Generator().manual_seed(8)).to(torch.int32)
            g = torch.where(nb.unsqueeze(-1) >= 0, a.double()[nb.clamp(min=0).long()], torch.zeros(1, dtype=torch.float64))   # [rows, 9, cin]
            want = torch.relu(torch.einsum("rtc,otc->ro", g, w.double()) + sh.double() + res.double())
            a_rows = nb.to(DEV)
        out = torch.zeros(rows, cout, device=DEV)
        _o.conv_rows(a.to(DEV), w.to(DEV), None, sh.to(DEV), out, a_rows=a_rows, taps=taps, m_cap=rows, relu=1, residual2d=res.to(DEV))
        err = (out.cpu().double() - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), err
    finally:
        _o.USE_DENSE_F32 = before[0]
        _o.set_math_mode(before[1])

