"""GPU parity tests of the fused spatial masker: the one-launch list build (ldn_mask_plan: ticketed prefix over the
images, patch-major lists, decisions from pooled means) and conv3's pooled-mean epilogue (ldn_conv_rows_pool), against the oracle's
index lists (models/utils.py:47-89, laud_resnet.py:96-110), the stand-alone kernels and plain fp64 arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fill import seeded_bernoulli, seeded_randn
from oracle import index_ref as IR

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from laudnet_amd import ops as _ops, load_library
    load_library()  # raises if libldn_hip.so is missing -- no fallback
    return _ops


@pytest.fixture(autouse=True)
def _mode():
    from laudnet_amd import ops as _ops
    _ops.set_math_mode("bf16x3")
    yield
    _ops.set_math_mode("fp32")


def _patch_major_order(m3, S, Sx):
    """The oracle's kept pixels re-ordered patch by patch (row-major inside a patch, patches row-major, images in order)."""
    B, Ho, Wo = m3.shape
    gy, gx = Ho // S, Wo // Sx
    rows, pre = [], [0]
    for b in range(B):
        for py in range(S):
            for px in range(Sx):
                for ly in range(gy):
                    for lx in range(gx):
                        y, x = py * gy + ly, px * gx + lx
                        if m3[b, y, x]:
                            rows.append((b * Ho + y) * Wo + x)
        pre.append(len(rows))
    return np.asarray(rows, dtype=np.int64), np.asarray(pre, dtype=np.int64)


def _check_patch_major(ix, patch, Ho, Wo, stride):
    B, S, Sx = patch.shape
    m3 = IR.upsample_patch_mask(patch.numpy() > 0.5, Ho, Wo)
    m1 = IR.dilate_mask(m3, stride, 1)
    want3, pre3 = _patch_major_order(m3, S, Sx)
    idx1, pre1 = IR.nonzero_rows(m1)
    cnt = ix.cnt.cpu().numpy()
    assert cnt[0] == len(want3) and cnt[1] == len(idx1)
    got3 = ix.idx3.cpu().numpy()[:cnt[0]]
    assert np.array_equal(got3, want3)                                  # bit-exact list, in the documented order
    assert np.array_equal(np.sort(got3), IR.nonzero_rows(m3)[0])        # ... the same SET as torch.nonzero's
    assert np.array_equal(ix.idx1.cpu().numpy()[:cnt[1]], idx1)
    assert np.array_equal(ix.pre3.cpu().numpy(), pre3) and np.array_equal(ix.pre1.cpu().numpy(), pre1)
    pos3 = np.full(B * Ho * Wo, -1, dtype=np.int64)
    pos3[want3] = np.arange(len(want3))
    assert np.array_equal(ix.pos3.cpu().numpy(), pos3)
    assert np.array_equal(ix.pos1.cpu().numpy(), IR.position_map(m1).reshape(-1))
    # the neighbour table of the row-major oracle, permuted to the patch-major row order
    raster = IR.nonzero_rows(m3)[0]
    nbr_raster = IR.neighbour_table(m3, m1, stride)
    where = {int(r): i for i, r in enumerate(raster)}
    want_nbr = nbr_raster[[where[int(r)] for r in want3]] if len(want3) else nbr_raster
    assert np.array_equal(ix.nbr.cpu().numpy().reshape(-1, 9)[:cnt[0]], want_nbr)
    want = np.array([patch.mean().item(), m3.mean(), m1.mean()], dtype=np.float32)
    assert np.allclose(ix.stats.cpu().numpy(), want, atol=1e-6)


@pytest.mark.parametrize("B,S,Ho,stride,p", [(3, 14, 56, 1, 0.5), (5, 7, 28, 1, 0.4), (256, 7, 14, 1, 0.5), (2, 7, 14, 2, 0.5),
                                             (4, 7, 28, 1, 0.0), (4, 7, 28, 1, 1.0), (300, 2, 8, 1, 0.5), (2, 4, 16, 2, 0.3),
                                             (1, 14, 56, 1, 0.5), (7, 7, 7, 1, 0.5)])
def test_plan_patch_major_lists(ops, B, S, Ho, stride, p):
    """ldn_mask_plan with a given mask, patch-major: bit-exact lists (the oracle's pixels, patch by patch), prefixes, positions,
    neighbour table and statistics -- B up to 300 images exercises the ticketed prefix over more workgroups than CUs."""
    patch = seeded_bernoulli((B, S, S), p, 3 + B + S + Ho)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride, patch_major=True)
    torch.cuda.synchronize()
    assert ix.patch_major
    _check_patch_major(ix, patch, Ho, Ho, stride)


def test_plan_nonsquare_patches_and_refusals(ops):
    import laudnet_amd._lib as L
    patch = seeded_bernoulli((3, 4, 6), 0.5, 17)
    ix = ops.mask_to_index(patch.to(DEV), 16, 12, 1, patch_major=True)     # 4 x 2 pixel patches
    torch.cuda.synchronize()
    _check_patch_major(ix, patch, 16, 12, 1)
    with pytest.raises(L.LdnError, match="even grid"):
        ops.mask_to_index(seeded_bernoulli((2, 3, 3), 0.5, 1).to(DEV), 14, 14, 1, patch_major=True)
    assert not ops.mask_plan_fits(25, 25, 200, 304, 1)                     # detection-size map: the banded build's job
    with pytest.raises(L.LdnError, match="exceed one workgroup"):
        ops.mask_to_index(seeded_bernoulli((1, 25, 19), 0.5, 1).to(DEV), 200, 304, 1, patch_major=True)


@pytest.mark.parametrize("B,S,Ho,Wo,stride,p", [(3, 14, 14, 14, 1, 0.5), (2, 3, 14, 14, 2, 0.4), (5, 14, 56, 56, 1, 0.3),
                                                (2, 9, 28, 20, 1, 0.5), (260, 7, 14, 14, 1, 0.5), (2, 4, 9, 23, 2, 0.5)])
def test_plan_equals_two_launch_build(ops, B, S, Ho, Wo, stride, p, monkeypatch):
    """The default ldn_mask_to_index (one launch, k_plan) against the two-launch build (k_mask_count + k_mask_index; LDN_INDEX_PLAN=0),
    list by list -- uneven grids and stride 2 included."""
    patch = seeded_bernoulli((B, S, S), p, 29 + B + Ho)
    one = ops.mask_to_index(patch.to(DEV), Ho, Wo, stride)
    monkeypatch.setenv("LDN_INDEX_PLAN", "0")
    two = ops.mask_to_index(patch.to(DEV), Ho, Wo, stride)
    torch.cuda.synchronize()
    n3, n1 = int(two.cnt[0]), int(two.cnt[1])
    assert torch.equal(one.cnt, two.cnt) and torch.equal(one.pre3, two.pre3) and torch.equal(one.pre1, two.pre1)
    assert torch.equal(one.stats, two.stats)
    assert torch.equal(one.pos3, two.pos3) and torch.equal(one.pos1, two.pos1)
    assert torch.equal(one.idx3[:n3], two.idx3[:n3]) and torch.equal(one.idx1[:n1], two.idx1[:n1])
    assert torch.equal(one.nbr[:n3 * 9], two.nbr[:n3 * 9])


def test_plan_leaves_its_flag_words_zero_for_the_next_build(ops):
    """ldn_plan_work_zeroed: the one-launch build clears its flag words on the way out (last workgroup out), so ONE buffer zeroed once per
    (device, stream) serves every build without a zeroing launch in front (ops' default).  A run of builds of different shapes on that
    buffer against builds with a fresh buffer + zeroing launch each (USE_CLEAN_PLAN_WORK = False): identical lists; the shared buffer is all
    zero afterwards."""
    shapes = [(260, 7, 14, 14, 1, 0.5), (3, 14, 14, 14, 1, 0.5), (5, 14, 56, 56, 1, 0.3), (2, 3, 14, 14, 2, 0.4), (64, 7, 28, 28, 1, 0.6), (260, 7, 14, 14, 1, 0.1)]
    assert ops.USE_CLEAN_PLAN_WORK
    got = []
    for rep in range(2):
        for (B, S, Ho, Wo, stride, p) in shapes:
            got.append(ops.mask_to_index(seeded_bernoulli((B, S, S), p, 3 + B + Ho).to(DEV), Ho, Wo, stride))
    torch.cuda.synchronize()
    bufs = [w for k, w in ops._PLAN_WORK.items() if k[0] == str(torch.device(DEV)) or k[0] == DEV]
    assert bufs and all(int(w.abs().sum()) == 0 for w in bufs), "the flag words must be left zero"
    ops.USE_CLEAN_PLAN_WORK = False
    try:
        want = [ops.mask_to_index(seeded_bernoulli((B, S, S), p, 3 + B + Ho).to(DEV), Ho, Wo, stride) for (B, S, Ho, Wo, stride, p) in shapes]
    finally:
        ops.USE_CLEAN_PLAN_WORK = True
    torch.cuda.synchronize()
    for i, one in enumerate(got):
        two = want[i % len(shapes)]
        n3, n1 = int(two.cnt[0]), int(two.cnt[1])
        assert torch.equal(one.cnt, two.cnt) and torch.equal(one.pre3, two.pre3) and torch.equal(one.pre1, two.pre1) and torch.equal(one.stats, two.stats)
        assert torch.equal(one.idx3[:n3], two.idx3[:n3]) and torch.equal(one.idx1[:n1], two.idx1[:n1]) and torch.equal(one.nbr[:n3 * 9], two.nbr[:n3 * 9])


@pytest.mark.parametrize("B,C,S,H", [(4, 256, 14, 56), (3, 512, 7, 28), (6, 1024, 7, 14), (2, 64, 4, 8)])
def test_plan_decides_like_the_standalone_masker(ops, B, C, S, H):
    """Decide mode: from the pooled means the stand-alone masker stored, ldn_mask_plan takes the SAME decisions with the SAME logits
    (bit-identical: one arithmetic order), and its lists are those of ldn_mask_to_index on that mask."""
    w = seeded_randn((2, C), 5).to(DEV) * 0.1
    b = torch.tensor([0.01, -0.02], device=DEV)
    x = F.relu(seeded_randn((B, H, H, C), 9)).to(DEV).contiguous()
    mask0, lg0, work = ops.spatial_masker(x, w, b, 1, S, want_logits=True, return_work=True)
    mask1, lg1, ix = ops.mask_plan(work.view(B, S, S, C), w, b, H, H, 1, patch_major=True, want_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(lg1, lg0) and torch.equal(mask1, mask0)
    assert 0.05 < float(mask0.mean()) < 0.95
    _check_patch_major(ix, mask0[:, 0].cpu(), H, H, 1)
    _, _, ixr = ops.mask_plan(work.view(B, S, S, C), w, b, H, H, 1, patch_major=False)
    ref = ops.mask_to_index(mask0[:, 0].contiguous(), H, H, 1)
    n3 = int(ref.cnt[0])
    assert torch.equal(ixr.cnt, ref.cnt) and torch.equal(ixr.idx3[:n3], ref.idx3[:n3]) and torch.equal(ixr.nbr[:9 * n3], ref.nbr[:9 * n3])


@pytest.mark.parametrize("B,S,H,cin,cout,p", [(3, 14, 56, 64, 256, 0.5), (4, 7, 28, 128, 512, 0.4), (5, 7, 14, 256, 1024, 0.5),
                                              (2, 7, 14, 64, 128, 1.0)])
def test_conv_rows_pool_epilogue(ops, math_mode, B, S, H, cin, cout, p):
    """conv3 with the pooled-mean by-product: the output equals the plain call's bit for bit (row order does not enter a row's
    arithmetic), the pool rows of the patches it wrote hold the mean of the final output over the patch (fp32 sum of 4 / 16 values
    against fp64: 1e-6 relative), the other rows are untouched."""
    ops.set_math_mode(math_mode)
    if math_mode == "fp32":
        ops.USE_DENSE_F32, keep = True, ops.USE_DENSE_F32
    try:
        patch = seeded_bernoulli((B, S, S), p, 41 + B + S)
        ix = ops.mask_to_index(patch.to(DEV), H, H, 1, patch_major=True)
        ixr = ops.mask_to_index(patch.to(DEV), H, H, 1)
        n = int(ix.cnt[0])
        h2 = seeded_randn((ix.cap3, cin), 3).to(DEV)
        w3 = (seeded_randn((cout, 1, cin), 4) * 0.1).to(DEV)
        t3 = seeded_randn((cout,), 5).to(DEV)
        x = F.relu(seeded_randn((B * H * H, cout), 6)).to(DEV)
        pool = torch.full((B, S, S, cout), -7.0, device=DEV)
        out = x.clone()
        ops.conv_rows(h2, w3, None, t3, out, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_rows=ix.idx3, residual2d=out,
                      pool=pool, pool_grid=(S, S, H, H))
        # the same rows in row-major order through the plain entry point: h2 rows permuted to follow the pixel they belong to
        perm = ixr.pos3[ix.idx3[:n].long()].long()          # patch-major row r holds the pixel whose row-major rank is perm[r]
        h2r = torch.zeros_like(h2)
        h2r[perm] = h2[:n]
        want = x.clone()
        ops.conv_rows(h2r, w3, None, t3, want, taps=1, m_count=ixr.cnt[0:1], m_cap=ixr.cap3, relu=1, out_rows=ixr.idx3, residual2d=want)
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        g = H // S
        means = out.double().view(B, S, g, S, g, cout).mean(dim=(2, 4))            # [B,S,S,cout]
        kept = patch.to(DEV) > 0.5
        assert torch.allclose(pool[kept].double(), means[kept], rtol=1e-6, atol=1e-6)
        assert bool((pool[~kept] == -7.0).all())
    finally:
        if math_mode == "fp32":
            ops.USE_DENSE_F32 = keep


def _spatial_model(gran, seed=3):
    import laudnet_amd
    from fill import fill_state_dict
    m = laudnet_amd.uni_resnet50(dyn_mode=["spatial"] * 4, mask_spatial_granularity=gran, width_mult=0.5, input_size=224, num_classes=10).eval()
    sd = fill_state_dict(m.state_dict(), seed)
    for k in sd:       # zero the keep bias: fresh maskers keep everything (bias 5.0), nothing would be decided
        if k.endswith("masker_spatial.conv.bias"):
            sd[k] = torch.zeros_like(sd[k])
    m.load_state_dict(sd)
    return m.to(DEV)


@pytest.mark.parametrize("gran", [[4, 4, 2, 1], [2, 2, 1, 1]])
def test_fused_spatial_masker_whole_model(ops, gran):
    """LAUD-ResNet50 spatial: with the fused masker (means from conv3's epilogue, one plan launch per block) every block takes the
    decisions of the stand-alone path and the logits agree bit for bit -- unless a decision sits on a tie of the two summation orders
    (then the test says so instead of passing silently)."""
    from laudnet_amd import laud_resnet as LR
    m = _spatial_model(gran)
    x = seeded_randn((4, 3, 224, 224), 21).to(DEV)
    blocks = [b for i in range(4) for b in getattr(m, f"layer{i + 1}")]
    outs = {}
    for fused in (False, True):
        LR.Bottleneck.use_fused_spatial_masker = fused
        try:
            with torch.no_grad():
                logits = m(x, 1.0)[0]
            outs[fused] = (logits.clone(), [b.last_spatial_mask.clone() for b in blocks],
                           [bool(getattr(b, "last_carry", None) and len(b.last_carry) > 4 and b.last_carry[4]) for b in blocks])
        finally:
            LR.Bottleneck.use_fused_spatial_masker = True
    torch.cuda.synchronize()
    assert sum(outs[True][2]) >= 3, "the fused path must actually run on the identity blocks with 4- / 16-pixel patches"
    assert not any(outs[False][2])
    flips = [i for i, (a, b) in enumerate(zip(outs[False][1], outs[True][1])) if not torch.equal(a, b)]
    assert not flips, f"decisions differ at blocks {flips} (a tie between the two summation orders of the pooled means?)"
    assert torch.equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize("mode", ["spatial", "layer"])
def test_fused_masker_across_projection_blocks_and_stage_boundaries(ops, mode):
    """Round 5, late: (a) a stride-2 / projection block at the head of a stage decides from the cell means its predecessor (the last identity
    block of the stage before) left, when the two grids coincide; (b) a projection block LEAVES cell means itself -- its projection launch
    lists the output pixels cell by cell, so the block behind it decides without reading x either (models/utils.py:47-65 on the block INPUT).
    Which blocks decide from means is pinned (LAUD-ResNet50 @224: every block except the one behind the stem and the per-pixel / odd-map
    blocks of stage 4; spatial's stage-2 head averages the 2 x 2 groups of its predecessor's cells first); decisions and logits equal the
    stand-alone path's."""
    import laudnet_amd
    from laudnet_amd import laud_resnet as LR
    from fill import fill_state_dict
    kw = dict(mask_spatial_granularity=[4, 4, 2, 1]) if mode == "spatial" else {}
    m = laudnet_amd.uni_resnet50(dyn_mode=[mode] * 4, width_mult=0.5, input_size=224, num_classes=10, **kw).eval()
    sd = fill_state_dict(m.state_dict(), 7)
    for k in sd:
        if k.endswith("masker_spatial.conv.bias"):
            sd[k] = torch.zeros_like(sd[k])
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = seeded_randn((6, 3, 224, 224), 25).to(DEV)
    names = [f"{i + 1}.{j + 1}" for i in range(4) for j in range(len(getattr(m, f"layer{i + 1}")))]
    blocks = [b for i in range(4) for b in getattr(m, f"layer{i + 1}")]
    outs = {}
    for on in (True, False):
        LR.Bottleneck.use_fused_projection_means = on
        LR.ResNet.use_stage_carry = on
        try:
            with torch.no_grad():
                logits = m(x, 1.0)[0]
            outs[on] = (logits.clone(), [b.last_spatial_mask.clone() for b in blocks], [n for n, b in zip(names, blocks) if b.last_fused_decision])
        finally:
            LR.Bottleneck.use_fused_projection_means = True
            LR.ResNet.use_stage_carry = True
    torch.cuda.synchronize()
    every = set(names)
    expect_on = every - {"1.1", "4.2", "4.3"}      # (spatial 2.1: its 8 x 8-pixel cells are the 2 x 2 groups of block 1.3's -- ldn_coarsen_cell_means)
    expect_off = every - {"1.1", "1.2", "2.1", "2.2", "3.1", "3.2", "4.1", "4.2", "4.3"}
    assert set(outs[True][2]) == expect_on, sorted(every - set(outs[True][2]))
    assert set(outs[False][2]) == expect_off, sorted(every - set(outs[False][2]))
    flips = [names[i] for i, (a, b) in enumerate(zip(outs[False][1], outs[True][1])) if not torch.equal(a, b)]
    assert not flips, f"decisions differ at blocks {flips} (a tie between the two summation orders of the pooled means?)"
    assert torch.equal(outs[True][0], outs[False][0])


def test_fused_spatial_masker_graph_replay(ops):
    """The fused path inside a captured hipGraph (ticket words zeroed by a kernel node, decisions and list sizes on the device): three
    replays equal the eager forward bit for bit."""
    from laudnet_amd.laud_resnet import GraphedForward
    m = _spatial_model([4, 4, 2, 1])
    x = seeded_randn((8, 3, 224, 224), 23).to(DEV)
    with torch.no_grad():
        want = m(x, 1.0)[0].clone()
    assert any(len(b.last_carry or ()) > 4 and b.last_carry[4] for b in m.layer3)
    g = GraphedForward(m, x, 1.0)
    for _ in range(3):
        got = g(x)[0]
        torch.cuda.synchronize()
        assert torch.equal(got, want)


@pytest.mark.parametrize("B,Ho,stride,tile,p", [(9, 14, 1, (2, 2), 0.5), (256, 14, 1, (2, 2), 0.5), (5, 56, 1, (4, 4), 0.6), (7, 28, 2, (4, 4), 0.4),
                                                (4, 28, 1, (2, 2), 1.0), (4, 28, 1, (4, 4), 0.0), (3, 12, 1, (4, 2), 0.5)])
def test_layer_index_tile_order(ops, B, Ho, stride, tile, p):
    """ldn_layer_index: one decision per image, the kept images' pixels tile by tile -- bit-exact lists (the oracle's pixels in the
    documented order), and with tile = None the lists of ldn_mask_to_index."""
    patch = seeded_bernoulli((B, 1, 1), p, 13 + B + Ho)
    ix = ops.layer_index(patch.to(DEV).reshape(B), Ho, Ho, stride, tile=tile)
    torch.cuda.synchronize()
    # the oracle sees a patch grid of Ho/gy x Ho/gx cells that all carry the image's decision
    S, Sx = Ho // tile[0], Ho // tile[1]
    cells = patch.expand(B, S, Sx).contiguous()
    m3 = IR.upsample_patch_mask(cells.numpy() > 0.5, Ho, Ho)
    m1 = IR.dilate_mask(m3, stride, 1)
    want3, pre3 = _patch_major_order(m3, S, Sx)
    cnt = ix.cnt.cpu().numpy()
    assert cnt[0] == len(want3) and np.array_equal(ix.idx3.cpu().numpy()[:cnt[0]], want3)
    assert np.array_equal(ix.pre3.cpu().numpy(), pre3)
    pos3 = np.full(B * Ho * Ho, -1, dtype=np.int64)
    pos3[want3] = np.arange(len(want3))
    assert np.array_equal(ix.pos3.cpu().numpy(), pos3)
    idx1, pre1 = IR.nonzero_rows(m1)
    assert cnt[1] == len(idx1) and np.array_equal(ix.idx1.cpu().numpy()[:cnt[1]], idx1) and np.array_equal(ix.pre1.cpu().numpy(), pre1)
    raster = IR.nonzero_rows(m3)[0]
    where = {int(r): i for i, r in enumerate(raster)}
    nbr_raster = IR.neighbour_table(m3, m1, stride)
    want_nbr = nbr_raster[[where[int(r)] for r in want3]] if len(want3) else nbr_raster
    assert np.array_equal(ix.nbr.cpu().numpy().reshape(-1, 9)[:cnt[0]], want_nbr)
    assert np.allclose(ix.stats.cpu().numpy(), np.array([patch.mean().item(), m3.mean(), m1.mean()], dtype=np.float32), atol=1e-6)
    plain = ops.layer_index(patch.to(DEV).reshape(B), Ho, Ho, stride)
    ref = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    n3 = int(ref.cnt[0])
    assert torch.equal(plain.cnt, ref.cnt) and torch.equal(plain.idx3[:n3], ref.idx3[:n3]) and torch.equal(plain.pos3, ref.pos3)
    assert torch.equal(plain.nbr[:9 * n3], ref.nbr[:9 * n3])


def test_layer_head_from_tile_means(ops):
    """The layer-skip decision from tile means: the mean of the means of equal tiles is the global average pool -- logits within fp32
    reduction-order noise of the stand-alone layer masker's, decisions equal away from ties."""
    B, C, H = 6, 256, 28
    w = seeded_randn((2, C), 5).to(DEV) * 0.1
    b = torch.tensor([0.0, 0.0], device=DEV)
    x = F.relu(seeded_randn((B, H, H, C), 9)).to(DEV).contiguous()
    mask0, lg0 = ops.spatial_masker(x, w, b, 1, 1, want_logits=True)
    _, _, work = ops.spatial_masker(x, w, b, 1, 7, return_work=True)
    mask1, lg1 = ops.layer_head(work.view(B, 49, C), w, b, 1, want_logits=True)
    torch.cuda.synchronize()
    assert torch.allclose(lg1.reshape(-1), lg0.reshape(-1), atol=1e-5, rtol=1e-5)
    margin = (lg0.reshape(B, 2)[:, 0] - lg0.reshape(B, 2)[:, 1]).abs()
    differs = mask1.reshape(B) != mask0.reshape(B)
    assert not bool((differs & (margin > 1e-4)).any())


def test_fused_layer_masker_whole_model(ops):
    """LAUD-ResNet50 layer skip: with the fused masker (tile means from conv3's epilogue, the decision from them) every block takes the
    decisions of the stand-alone path and the logits agree bit for bit (a decision on a tie of the two summation orders would be named)."""
    import laudnet_amd
    from laudnet_amd import laud_resnet as LR
    from fill import fill_state_dict
    m = laudnet_amd.uni_resnet50(dyn_mode=["layer"] * 4, width_mult=0.5, input_size=224, num_classes=10).eval()
    sd = fill_state_dict(m.state_dict(), 3)
    for k in sd:
        if k.endswith("masker_spatial.conv.bias"):
            sd[k] = torch.zeros_like(sd[k])
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = seeded_randn((12, 3, 224, 224), 21).to(DEV)
    blocks = [b for i in range(4) for b in getattr(m, f"layer{i + 1}")]
    outs = {}
    for fused in (False, True):
        LR.Bottleneck.use_fused_spatial_masker = fused
        try:
            with torch.no_grad():
                logits = m(x, 1.0)[0]
            outs[fused] = (logits.clone(), [b.last_spatial_mask.clone() for b in blocks],
                           [bool(getattr(b, "last_carry", None) and len(b.last_carry) > 4 and b.last_carry[4]) for b in blocks])
        finally:
            LR.Bottleneck.use_fused_spatial_masker = True
    torch.cuda.synchronize()
    assert sum(outs[True][2]) >= 6 and not any(outs[False][2])
    kept = torch.cat([a.reshape(-1) for a in outs[False][1]])
    assert 0.05 < float(kept.mean()) < 0.95
    flips = [i for i, (a, b) in enumerate(zip(outs[False][1], outs[True][1])) if not torch.equal(a, b)]
    assert not flips, f"decisions differ at blocks {flips} (a tie between the two summation orders of the pooled means?)"
    assert torch.equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize("rows,cin,cout", [(25000, 384, 384), (42000, 1024, 256), (6000, 512, 512)])
def test_tile_width_hint_never_changes_results(ops, rows, cin, cout):
    """ldn_hint_rows (ops.conv_rows(rows_hint=...)) selects the tile width of k_dense from a cost model -- 64 / 128 / 192 /
    256-column tiles here; every output element's K loop is the same in every shape: bit-identical outputs whatever the hint says
    (missing, exact, far too small, far too large)."""
    cap = 2 * rows
    a = seeded_randn((cap, cin), 3).to(DEV)
    w = (seeded_randn((cout, 1, cin), 4) * 0.05).to(DEV)
    t = seeded_randn((cout,), 5).to(DEV)
    cnt = torch.tensor([rows], dtype=torch.int32, device=DEV)
    outs = []
    for hint in (None, rows, 100, cap, 3 * rows // 2):
        out = torch.zeros(cap, cout, device=DEV)
        ops.conv_rows(a, w, None, t, out, taps=1, m_count=cnt, m_cap=cap, relu=1, rows_hint=hint)
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert float(outs[0][rows:].abs().max()) == 0.0 and float(outs[0][:rows].abs().max()) > 0.0
    want = torch.relu(a[:64].double() @ w[:, 0].double().t() + t.double())
    assert torch.allclose(outs[0][:64].double(), want, atol=2e-4, rtol=1e-4)
