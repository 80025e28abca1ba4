"""-m gpu: the pre-split packed path of round 5 (ldn_conv_rows_ps, ldn_conv3x3_rows_ps -- k_dense<.., PS / OF> and k_rows3), through the C ABI.

A spatial / layer block keeps h1 and h2 PRE-SPLIT (bf16 hi | lo per octet of channels) between conv1, the packed 3x3 and conv3, so that no
operand is split inside a K loop (models/laud_resnet.py:115-144 on the active pixels; DyNetSimulator/eval_example.py:39-48).  The split
values are the ones the in-loop split of round 4's kernels computes and the K order is the same, so every result must be BIT-IDENTICAL to
the un-split kernels' (ldn_conv_rows_split) -- asserted here with torch.equal -- and within 1e-4 of an fp64 reference."""
import pytest
import torch

from fill import seeded_bernoulli, seeded_randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _bf16x3():
    from laudnet_amd import ops
    ops.set_math_mode("bf16x3")
    taps = ops.DENSE_TAPS
    ops.DENSE_TAPS = (1, 9)          # the bit-exact reference of the 3x3 is k_dense's neighbour-table form, not round 1's producer / consumer kernel
    yield
    ops.DENSE_TAPS = taps
    ops.set_math_mode("fp32")


@pytest.fixture(scope="module")
def ops():
    from laudnet_amd import ops as _ops, load_library
    load_library()
    return _ops


def _affine(c, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.1).to(DEV)


def test_presplit_helpers_round_trip(ops):
    x = seeded_randn((37, 64), 1).to(DEV)
    ps = ops.presplit_rows(x)
    assert ps.shape == x.shape and ps.dtype == torch.float32
    back = ops.unsplit_rows(ps)
    assert (back - x).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()


@pytest.mark.parametrize("rows,cin,cout,count", [(1000, 64, 64, 1000), (700, 256, 128, 613), (300, 1024, 256, 300), (513, 512, 512, 1)])
def test_conv1_writes_presplit_rows(ops, rows, cin, cout, count):
    """conv1's epilogue in the OF form: the pre-split rows it stores are exactly the bf16 hi / lo split of what the fp32 form stores."""
    x = seeded_randn((rows + 50, cin), 2).to(DEV)
    gather = torch.randperm(rows + 50, generator=torch.Generator().manual_seed(3))[:rows].to(torch.int32).to(DEV)
    w = (seeded_randn((cout, 1, cin), 4) * (2.0 / cin) ** 0.5).to(DEV)
    sc, sh = _affine(cout, 5)
    cnt = torch.tensor([count], dtype=torch.int32, device=DEV)
    want = torch.zeros(rows, cout, device=DEV)
    ops.conv_rows(x, w, sc, sh, want, a_rows=gather, taps=1, m_count=cnt, m_cap=rows, relu=1)
    got = torch.zeros(rows, cout, device=DEV)
    ops.conv_rows_ps(x, w, sc, sh, got, out_presplit=True, a_rows=gather, m_count=cnt, m_cap=rows, relu=1)
    torch.cuda.synchronize()
    assert torch.equal(got[:count].view(torch.int32), ops.presplit_rows(want[:count]).view(torch.int32))
    assert torch.equal(got[count:], torch.zeros_like(got[count:])), "rows past the device-side count must not be written"
    ref = torch.relu((x[gather[:count].long()].double() @ w[:, 0].double().t()) * sc.double() + sh.double())
    assert (ops.unsplit_rows(got[:count]).double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,stride,C,cout,p", [(3, 14, 1, 64, 64, 0.5), (2, 14, 1, 256, 256, 0.5), (2, 28, 2, 128, 128, 0.6), (5, 7, 1, 128, 64, 1.0),
                                                 (1, 14, 1, 64, 128, 0.1), (40, 14, 1, 256, 256, 0.5)])
@pytest.mark.parametrize("out_ps", [False, True])
def test_rows3_bit_identical_to_unsplit_kernel(ops, B, H, stride, C, cout, p, out_ps):
    """k_rows3 (pre-split h1 through the neighbour table, weights in K64 steps, private row buffers) against k_dense<.., T9> on the same
    values: bit-identical; and against F.conv2d-style fp64 arithmetic on the gathered rows."""
    Ho = H // stride
    patch = seeded_bernoulli((B, Ho, Ho), p, 11)
    ix = ops.mask_to_index(patch.to(DEV), Ho, Ho, stride)
    n3 = int(ix.cnt[0])
    h1 = torch.relu(seeded_randn((ix.cap1, C), 12)).to(DEV)
    w = (seeded_randn((cout, 9, C), 13) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    sc, sh = _affine(cout, 14)
    want = torch.zeros(ix.cap3, cout, device=DEV)
    ops.conv_rows(h1, w, sc, sh, want, a_rows=ix.nbr, taps=9, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1)
    got = torch.zeros(ix.cap3, cout, device=DEV)
    for hint in (None, n3):        # both tile widths where the layer admits them (the hint only changes tile shapes)
        got.zero_()
        ops.conv3x3_rows_ps(ops.presplit_rows(h1), ix.nbr, w, sc, sh, got, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_presplit=out_ps, rows_hint=hint)
        torch.cuda.synchronize()
        if out_ps:
            assert torch.equal(got[:n3].view(torch.int32), ops.presplit_rows(want[:n3]).view(torch.int32)), hint
        else:
            assert torch.equal(got[:n3], want[:n3]), hint
        assert torch.equal(got[n3:], torch.zeros_like(got[n3:])), "rows past the device-side count must not be written"
    # fp64 reference on the gathered neighbour rows
    nbr = ix.nbr.view(-1, 9)[:n3].long()
    h1z = torch.cat((h1.double(), torch.zeros(1, C, dtype=torch.float64, device=DEV)))
    gath = h1z[torch.where(nbr >= 0, nbr, torch.full_like(nbr, ix.cap1))]                    # [n3, 9, C]
    ref = torch.relu(torch.einsum("mtk,ntk->mn", gath, w.double()) * sc.double() + sh.double())
    val = ops.unsplit_rows(got[:n3]) if out_ps else got[:n3]
    assert (val.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("rows,cin,cout,count", [(900, 64, 256, 900), (700, 256, 1024, 650), (257, 128, 512, 257), (400, 512, 2048, 33)])
def test_conv3_reads_presplit_rows(ops, rows, cin, cout, count):
    """conv3 in the PS form (scatter-add into the residual stream + ReLU): bit-identical to the form that splits h2 in its K loop."""
    h2 = torch.relu(seeded_randn((rows, cin), 21)).to(DEV)
    w = (seeded_randn((cout, 1, cin), 22) * (2.0 / cin) ** 0.5).to(DEV)
    _, sh = _affine(cout, 23)
    dst = torch.randperm(rows * 2, generator=torch.Generator().manual_seed(24))[:rows].to(torch.int32).to(DEV)
    cnt = torch.tensor([count], dtype=torch.int32, device=DEV)
    base = torch.relu(seeded_randn((rows * 2, cout), 25)).to(DEV)
    want = base.clone()
    ops.conv_rows(h2, w, None, sh, want, taps=1, m_count=cnt, m_cap=rows, relu=1, out_rows=dst, residual2d=want)
    got = base.clone()
    ops.conv_rows_ps(ops.presplit_rows(h2), w, None, sh, got, a_presplit=True, m_count=cnt, m_cap=rows, relu=1, out_rows=dst, residual2d=got)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = base.double().clone()
    d = dst[:count].long()
    ref[d] = torch.relu(ref[d] + h2[:count].double() @ w[:, 0].double().t() + sh.double())
    assert (got.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_conv3_presplit_leaves_the_pooled_patch_means(ops):
    """The fused spatial masker's by-product (ldn_conv_rows_pool) on the PS form: same pooled means as the un-split form."""
    B, H, S, cin, cout = 6, 28, 7, 128, 512
    patch = seeded_bernoulli((B, S, S), 0.5, 31)
    ix = ops.mask_to_index(patch.to(DEV), H, H, 1, patch_major=True)
    n3 = int(ix.cnt[0])
    h2 = torch.relu(seeded_randn((ix.cap3, cin), 32)).to(DEV)
    w = (seeded_randn((cout, 1, cin), 33) * (2.0 / cin) ** 0.5).to(DEV)
    _, sh = _affine(cout, 34)
    x = torch.relu(seeded_randn((B * H * H, cout), 35)).to(DEV)
    outs, pools = [], []
    for ps in (False, True):
        out = x.clone()
        pool = torch.zeros(B, S, S, cout, device=DEV)
        if ps:
            ops.conv_rows_ps(ops.presplit_rows(h2), w, None, sh, out, a_presplit=True, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_rows=ix.idx3,
                             residual2d=out, pool=pool, pool_grid=(S, S, H, H))
        else:
            ops.conv_rows(h2, w, None, sh, out, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1, out_rows=ix.idx3, residual2d=out, pool=pool,
                          pool_grid=(S, S, H, H))
        outs.append(out)
        pools.append(pool)
    torch.cuda.synchronize()
    assert n3 > 0 and torch.equal(outs[0], outs[1]) and torch.equal(pools[0], pools[1])


@pytest.mark.parametrize("mode", ["spatial", "layer"])
def test_block_presplit_path_bit_identical_to_round4_path(ops, mode):
    """A whole spatial / layer bottleneck at real widths: the pre-split path (default) against the three un-split launches (LDN_ROWS_PS=0
    semantics, switched through ops.USE_ROWS_PS): identical masks, bit-identical outputs."""
    import laudnet_amd
    from laudnet_amd.laud_resnet import Bottleneck
    from fill import fill_state_dict
    torch.manual_seed(0)
    blk = Bottleneck(1024, 256, stride=1, dyn_mode=mode, output_size=14, mask_spatial_granularity=2).eval()
    blk.load_state_dict(fill_state_dict(blk.state_dict(), 41))
    blk = blk.to(DEV)
    x = torch.relu(seeded_randn((24, 1024, 14, 14), 42)).to(DEV).contiguous(memory_format=torch.channels_last)
    ms = blk.masker_spatial
    blk.forced_spatial_mask = seeded_bernoulli((24, 1, ms.mask_size, ms.mask_size), 0.5, 43).to(DEV)
    outs = []
    for flag in (True, False):
        ops.USE_ROWS_PS = flag
        try:
            with torch.no_grad():
                y, _ = blk.run_dynamic(x)
            outs.append(y.clone())
        finally:
            ops.USE_ROWS_PS = True
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
