"""CPU checks of the one-time weight / activation layouts the fused kernels read (laudnet_amd/ops.py packers; include/ldn_hip.h):
the bf16 hi | lo ("pre-split") forms and their fp32 twins hold the same values at the same positions -- an element is 4 bytes either
way, which is what lets k_head / k_tail / k_chain / k_dense share their staging between the two arithmetic modes."""
import torch

from fill import seeded_randn


def test_w2_pairs_bf16_and_fp32_twins_agree():
    from laudnet_amd import ops
    W = 64
    w = seeded_randn((W, W, 3, 3), 1) * 0.1
    pb = ops.pack_w2_pairs(w)                 # [9][kp][np][nn][hi|lo][kk] bf16
    pf = ops.pack_w2_pairs(w, f32=True)       # [9][kp][np][nn][kk] fp32
    assert pb.dtype == torch.bfloat16 and pf.dtype == torch.float32
    assert tuple(pb.shape) == (9, W // 2, W // 2, 2, 2, 2) and tuple(pf.shape) == (9, W // 2, W // 2, 2, 2)
    assert pb.numel() * 2 == pf.numel() * 4                                     # same bytes
    dec = pb.float().sum(dim=-2)                                                # hi + lo
    assert torch.allclose(dec, pf, atol=0.0, rtol=2e-5)
    # entry (tap, kp, np, nn, kk) = conv2.weight[n = 2 np + nn, k = 2 kp + kk, tap]
    for (t, kp, np_, nn, kk) in [(0, 0, 0, 0, 0), (4, 3, 7, 1, 0), (8, 31, 31, 1, 1), (5, 10, 2, 0, 1)]:
        assert pf[t, kp, np_, nn, kk].item() == w[2 * np_ + nn, 2 * kp + kk, t // 3, t % 3].item()


def test_w3_pairs_and_w1_twins_agree():
    from laudnet_amd import ops
    cout, W, cin = 256, 64, 128
    w3 = seeded_randn((cout, W), 2) * 0.1
    pb, pf = ops.pack_w3_pairs(w3), ops.pack_w3_pairs(w3, f32=True)
    assert tuple(pb.shape) == (W // 2, cout, 2, 2) and tuple(pf.shape) == (W // 2, cout, 2)
    assert torch.allclose(pb.float().sum(dim=-2), pf, atol=0.0, rtol=2e-5)
    assert pf[5, 17, 1].item() == w3[17, 11].item()                             # entry (kp, c, kk) = w3[c, 2 kp + kk]
    w1 = seeded_randn((W, cin), 3) * 0.1
    sb, sf = ops.pack_w1_split(w1), ops.pack_w1_split(w1, f32=True)
    assert tuple(sb.shape) == (W, cin // 8, 2, 8) and tuple(sf.shape) == (W, cin)
    assert torch.allclose(sb.float().sum(dim=-2).reshape(W, cin), sf, atol=0.0, rtol=2e-5)
    assert torch.equal(sf, w1)                                                  # [n][octet][8 floats] IS row-major fp32


def test_x_split_tile_layout_roundtrip():
    """decode_x_split against an independent encoder of the 32-pixel tile layout of ldn_bottleneck_head_split:
    [tile][K16 step s][octet h of the step][hi | lo][pixel % 32][8 bf16]; channel = 16 s + 8 h + i."""
    from laudnet_amd import ops
    pixels, cin = 75, 64                       # a ragged last tile
    x = seeded_randn((pixels, cin), 4)
    ntile = (pixels + 31) // 32
    buf = torch.zeros(ntile, cin // 16, 2, 2, 32, 8, dtype=torch.bfloat16)
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    for q in range(pixels):
        for s in range(cin // 16):
            for h in range(2):
                c0 = 16 * s + 8 * h
                buf[q // 32, s, h, 0, q % 32] = hi[q, c0:c0 + 8]
                buf[q // 32, s, h, 1, q % 32] = lo[q, c0:c0 + 8]
    got = ops.decode_x_split(buf.view(torch.float32).reshape(-1), pixels, cin)
    assert torch.allclose(got, x, atol=0.0, rtol=2e-5)
