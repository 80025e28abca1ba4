"""-m gpu: the SE block of a LAD-RegNet layer-skip bottleneck folded into its neighbours (round 5), through the C ABI.

Reference semantics: laud_regnet.py:194-197 `x = self.b(x); x = self.se(x); x = self.c(x)` with torchvision's SqueezeExcitation
(laud_regnet.py:119-123: scale = sigmoid(fc2(relu(fc1(avgpool(x))))); x * scale).  Unfused the library runs conv b, a channel-sum pass, the
SE head, an in-place scaling pass and conv c; fused, conv b's epilogue leaves the channel sums (ldn_grouped16_conv3x3_images_gap), the head
makes a gate per kept image (ldn_se_gate_slots) and conv c multiplies its input rows by it in flight (ldn_conv_rows_gated)."""
import pytest
import torch

from fill import seeded_bernoulli, seeded_randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _bf16x3():
    from laudnet_amd import ops
    ops.set_math_mode("bf16x3")
    yield
    ops.set_math_mode("fp32")


@pytest.fixture(scope="module")
def ops():
    from laudnet_amd import ops as _ops, load_library
    load_library()
    return _ops


def _affine(c, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.1).to(DEV)


@pytest.mark.parametrize("B,Ho,stride,C,p", [(6, 14, 1, 320, 0.5), (4, 7, 1, 784, 0.5), (5, 14, 2, 320, 0.4), (3, 28, 1, 144, 0.6), (2, 5, 1, 16, 1.0),
                                             (4, 9, 2, 48, 0.0), (3, 56, 2, 64, 0.7), (2, 56, 1, 32, 1.0)])
def test_conv_b_leaves_the_channel_sums_of_its_output(ops, B, Ho, stride, C, p):
    """Same output rows as the plain whole-image kernel (bit-identical); gap[k, :, c].sum(bands) = the sum of output channel c over image k's pixels
    (fp64 reference of the sums of the kernel's own output); repeated launches give the same bits (fixed order of additions)."""
    Hi = Ho * stride
    patch = seeded_bernoulli((B, 1, 1), p, 31 + C + B)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    n3 = int(ix.cnt[0])
    kept = n3 // (Ho * Ho)
    h_a = torch.relu(seeded_randn((ix.cap1, C), 32 + C)).to(DEV)
    w = (seeded_randn((C, 9, 16), 33 + C) * (2.0 / 144) ** 0.5).to(DEV)
    frag = ops.pack_grouped16_weights(w)
    sc, sh = _affine(C, 34)
    images = (B, Hi, Hi, Ho, Ho, stride)
    want = torch.full((ix.cap3, C), -7.0, device=DEV)
    ops.grouped16_conv3x3_images(h_a, frag, sc, sh, want, m_count=ix.cnt[0:1], images=images, relu=1)
    got = torch.full((ix.cap3, C), -7.0, device=DEV)
    gap = ops.grouped16_conv3x3_images_gap(h_a, frag, sc, sh, got, m_count=ix.cnt[0:1], images=images, relu=1)
    gap2 = ops.grouped16_conv3x3_images_gap(h_a, frag, sc, sh, got, m_count=ix.cnt[0:1], images=images, relu=1)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    bands = ops.grouped16_images_bands(Hi, Hi, Ho, stride, C)
    assert gap.shape == (B, bands, C) and bands >= 1
    assert torch.equal(gap[:kept], gap2[:kept])
    if kept:
        ref = got[:n3].double().view(kept, Ho * Ho, C).sum(dim=1)
        assert (gap[:kept].double().sum(dim=1) - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,kept,bands,C,S,rows", [(8, 5, 1, 320, 80, 196), (4, 4, 3, 64, 8, 3136), (6, 0, 1, 784, 196, 49), (3, 1, 2, 144, 36, 784)])
def test_se_gate_of_the_kept_images(ops, B, kept, bands, C, S, rows):
    """gate[k] = sigmoid(W2 relu(W1 mean + b1) + b2) for the kept images only (torchvision SqueezeExcitation on the channel means)."""
    gap = (torch.rand(B, bands, C, generator=torch.Generator().manual_seed(1)) * rows / bands).to(DEV)
    w1 = (seeded_randn((S, C), 2) * (1.0 / C) ** 0.5).to(DEV)
    b1 = (seeded_randn((S,), 3) * 0.1).to(DEV)
    w2 = (seeded_randn((C, S), 4) * (1.0 / S) ** 0.5).to(DEV)
    b2 = (seeded_randn((C,), 5) * 0.1).to(DEV)
    cnt = torch.tensor([kept * rows], dtype=torch.int32, device=DEV)
    gate = ops.se_gate_slots(gap, cnt, rows, w1, b1, w2, b2)
    torch.cuda.synchronize()
    assert gate.shape == (B, C)
    if kept:
        mean = gap[:kept].double().sum(dim=1) / rows
        ref = torch.sigmoid(torch.relu(mean @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
        assert (gate[:kept].double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("imgs,kept,rows,cin,cout", [(6, 4, 196, 320, 320), (9, 7, 49, 784, 784), (3, 3, 784, 144, 144), (2, 2, 3136, 64, 64),
                                                     (5, 3, 196, 320, 784), (4, 4, 100, 72, 256), (3, 2, 49, 48, 100)])
def test_conv_c_with_the_gate_on_its_input_rows(ops, imgs, kept, rows, cin, cout):
    """ldn_conv_rows_gated (scatter-add into the residual stream + ReLU) against scaling h_b first and the plain kernel: bit-identical
    (the products are formed from the same fp32 values in the same K order); and against fp64."""
    h_b = torch.relu(seeded_randn((imgs * rows, cin), 11)).to(DEV)
    gate = torch.sigmoid(seeded_randn((imgs, cin), 12)).to(DEV)
    w = (seeded_randn((cout, 1, cin), 13) * (2.0 / cin) ** 0.5).to(DEV)
    sc, sh = _affine(cout, 14)
    n = kept * rows
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    dst = torch.randperm(imgs * rows * 2, generator=torch.Generator().manual_seed(15))[:imgs * rows].to(torch.int32).to(DEV)
    base = torch.relu(seeded_randn((imgs * rows * 2, cout), 16)).to(DEV)
    scaled = (h_b.view(imgs, rows, cin) * gate.view(imgs, 1, cin)).reshape(imgs * rows, cin).contiguous()
    want = base.clone()
    ops.conv_rows(scaled, w, sc, sh, want, taps=1, m_count=cnt, m_cap=imgs * rows, relu=1, out_rows=dst, residual2d=want)
    got = base.clone()
    ops.conv_rows_gated(h_b, w, sc, sh, got, gate, rows, m_count=cnt, m_cap=imgs * rows, relu=1, out_rows=dst, residual2d=got)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = base.double().clone()
    d = dst[:n].long()
    ref[d] = torch.relu(ref[d] + (scaled[:n].double() @ w[:, 0].double().t()) * sc.double() + sh.double())
    assert (got.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("rows_total,gate_rows,cin,cout,count", [(1000, 7, 8, 4, 1000), (513, 2, 40, 36, 400), (300, 300, 2048, 64, 257), (257, 64, 136, 200, 1),
                                                                 (2049, 49, 784, 320, 2049), (64, 10, 24, 160, 0)])
def test_conv_c_gated_edge_shapes(ops, rows_total, gate_rows, cin, cout, count):
    """Ragged everything: rows not a multiple of the tile, one gate row per input row, K tails (cin % 32 != 0), the widest K, column tails,
    an empty and a one-row device-side count -- against fp64 and bit-identical to scaling first."""
    imgs = (rows_total + gate_rows - 1) // gate_rows
    h_b = seeded_randn((rows_total, cin), 71).to(DEV)
    gate = torch.sigmoid(seeded_randn((imgs, cin), 72)).to(DEV)
    w = (seeded_randn((cout, 1, cin), 73) * (2.0 / cin) ** 0.5).to(DEV)
    sc, sh = _affine(cout, 74)
    cnt = torch.tensor([count], dtype=torch.int32, device=DEV)
    img_of_row = torch.arange(rows_total, device=DEV) // gate_rows
    scaled = (h_b * gate[img_of_row]).contiguous()
    want = torch.full((rows_total, cout), -3.0, device=DEV)
    ops.conv_rows(scaled, w, sc, sh, want, taps=1, m_count=cnt, m_cap=rows_total, relu=0)
    got = torch.full((rows_total, cout), -3.0, device=DEV)
    ops.conv_rows_gated(h_b, w, sc, sh, got, gate, gate_rows, m_count=cnt, m_cap=rows_total, relu=0)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert bool((got[count:] == -3.0).all()), "rows past the device-side count must not be written"
    if count:
        ref = (scaled[:count].double() @ w[:, 0].double().t()) * sc.double() + sh.double()
        assert (got[:count].double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_conv_c_gated_rejects_what_it_does_not_run(ops):
    from laudnet_amd import LdnError
    h_b = torch.zeros(64, 32, device=DEV)
    w = torch.zeros(32, 1, 32, device=DEV)
    sh = torch.zeros(32, device=DEV)
    out = torch.zeros(64, 32, device=DEV)
    with pytest.raises(LdnError):
        ops.conv_rows_gated(h_b, w, None, sh, out, torch.ones(1, 32, device=DEV), 32)        # gate covers one image, the rows span two
    with pytest.raises(LdnError):
        ops.conv_rows_gated(h_b, w, None, sh, out, torch.ones(2, 16, device=DEV), 32)        # gate width != cin
    with pytest.raises(LdnError):     # one gate row per input row of a 1024-wide layer: 257 gate vectors per tile do not fit the LDS
        ops.conv_rows_gated(torch.zeros(64, 1024, device=DEV), torch.zeros(32, 1, 1024, device=DEV), None, sh, out, torch.ones(64, 1024, device=DEV), 1)


@pytest.mark.parametrize("stride,Hi,w_in,w_out", [(1, 14, 320, 320), (2, 28, 144, 320), (1, 7, 784, 784)])
def test_block_with_folded_se_matches_the_six_launch_path(ops, stride, Hi, w_in, w_out):
    """A whole layer-skip ResBottleneckBlock at RegNetY-800MF widths: the folded SE (default) against the separate SE passes
    (LDN_SE_FUSED=0 semantics): same skip decisions; outputs equal to fp32 rounding of the two orders of the channel sums."""
    import torch.nn as nn
    from functools import partial
    from laudnet_amd import laud_regnet
    from fill import fill_state_dict
    blk = laud_regnet.ResBottleneckBlock(w_in, w_out, stride, partial(nn.BatchNorm2d), nn.ReLU, group_width=16, bottleneck_multiplier=1.0,
                                         se_ratio=0.25, output_size=Hi // stride, mask_spatial_granularity=Hi // stride, dyn_mode="spatial").eval()
    blk.load_state_dict(fill_state_dict(blk.state_dict(), 51))
    blk = blk.to(DEV)
    B = 12
    x = torch.relu(seeded_randn((B, w_in, Hi, Hi), 52)).to(DEV).contiguous(memory_format=torch.channels_last)
    blk.f.forced_spatial_mask = seeded_bernoulli((B, 1, 1, 1), 0.6, 53).to(DEV)
    outs = []
    for flag in (True, False):
        laud_regnet.USE_SE_FUSED = flag
        try:
            with torch.no_grad():
                y, _ = blk.run_dynamic(x)
            outs.append(y.clone())
        finally:
            laud_regnet.USE_SE_FUSED = True
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert (outs[0] - outs[1]).abs().max().item() < 1e-5 * max(1.0, outs[1].abs().max().item())
