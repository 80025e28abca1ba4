"""GPU parity of the one-launch small-map bottleneck (ldn_bottleneck_smallmap: conv1 -> conv2 3x3 -> conv3 + residual of a channel-mode
block on a map of at most 64 pixels, h1 / h2 in LDS, bf16x3 arithmetic) against the dense-emulation algebra of the reference
(laud_resnet.py:115-144, channel mask applied before BN).  Tolerance 2e-4 + 1e-4 relative on O(1) activations (north star: 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

from fill import seeded_bernoulli, seeded_randn
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from laudnet_amd import ops as _ops, load_library
    load_library()
    return _ops


@pytest.fixture(autouse=True)
def _bf16x3_mode():
    """k_smallmap is a bf16x3 kernel and Bottleneck.tail_weights follows the arithmetic mode (round 4: fp32 twins of the layouts)."""
    from laudnet_amd import ops as _ops
    _ops.set_math_mode("bf16x3")
    yield
    _ops.set_math_mode("fp32")


def _block_pair(cin, W, gran, Ho):
    from laudnet_amd.laud_resnet import Bottleneck
    blk = TR.BottleneckRef(cin, W, stride=1, downsample=None, dyn_mode="channel", channel_dyn_granularity=gran,
                           channel_masker="MLP", output_size=Ho).eval()
    TR.randomize_bn_(blk, 5)
    with torch.no_grad():
        for m in (blk.conv1, blk.conv2, blk.conv3):
            m.weight.normal_(0, (2.0 / (m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3])) ** 0.5)
    hb = Bottleneck(cin, W, stride=1, downsample=None, dyn_mode="channel", channel_dyn_granularity=gran,
                    channel_masker="MLP", output_size=Ho).eval()
    hb.load_state_dict(blk.state_dict())
    return blk, hb.to(DEV)


# (B, H, Wd, W, gran, keep): stage 4 of the ResNets (7x7, width 512), every ring shape (keep 1.0 at width 512: single-slot rings; keep
# 0.75: K16 chunks of conv2), maps of at most 32 pixels (the second pixel tile is empty), 64 pixels, non-square, small widths
CASES = [(3, 7, 7, 512, 2, 0.62), (3, 7, 7, 512, 2, 1.0), (3, 7, 7, 512, 4, 0.75), (2, 7, 7, 256, 2, 0.62), (3, 4, 4, 64, 2, 0.62),
         (2, 8, 8, 128, 4, 0.5), (2, 5, 9, 192, 2, 0.62), (9, 7, 7, 384, 2, 0.3)]


@pytest.mark.parametrize("B,H,Wd,W,gran,keep", CASES)
def test_smallmap_vs_reference_algebra(ops, B, H, Wd, W, gran, keep):
    G = W // gran
    cin = cout = 4 * W
    assert ops.bottleneck_smallmap_fits(H, Wd, cin, W, cout)
    gm = seeded_bernoulli((B, G), keep, 31 + H + W)
    gm[0] = 0.0          # an image with no active channel
    gm[1] = 1.0          # an image with all channels
    blk, hb = _block_pair(cin, W, gran, H)
    x = F.relu(seeded_randn((B, cin, H, Wd), 32))
    cm = TR.broadcast_channel_mask(gm, W)
    with torch.no_grad():
        h1 = F.relu(blk.bn1(blk.conv1(x) * cm))
        h2 = F.relu(blk.bn2(blk.conv2(h1) * cm))
        want = F.relu(blk.bn3(blk.conv3(h2)) + x).permute(0, 2, 3, 1)
    p = hb._prepare(torch.device(DEV))
    w2p, w3p = hb.tail_weights(p)
    _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm.to(DEV))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.full((B, H, Wd, cout), float("nan"), device=DEV)
    colsum = torch.full((B, 2, cout), float("nan"), device=DEV)
    ops.bottleneck_smallmap(xn, p["w1s"], w2p, w3p, idx, cnt, p["s1"], p["t1"], p["c1"], p["s2"], p["t2_tab"], p["c2"], p["t3c"], out,
                            residual=xn, colsum=colsum)
    torch.cuda.synchronize()
    err = (out.cpu() - want).abs()
    assert torch.allclose(out.cpu(), want, atol=2e-4, rtol=1e-4), f"max err {err.max().item():.3e} at {tuple(torch.nonzero(err == err.max())[0].tolist())}"
    assert torch.allclose(colsum.sum(dim=1).cpu().double(), out.cpu().double().sum(dim=(1, 2)), atol=1e-2, rtol=1e-5)
    # in-place residual stream (out aliases x and the residual): same result, bit for bit
    ops.bottleneck_smallmap(xn, p["w1s"], w2p, w3p, idx, cnt, p["s1"], p["t1"], p["c1"], p["s2"], p["t2_tab"], p["c2"], p["t3c"], xn,
                            residual=xn)
    assert torch.equal(xn, out)


def test_smallmap_matches_head_plus_tail(ops):
    """Same products and channel algebra as ldn_bottleneck_head + ldn_bottleneck_tail (widths both cover): agreement to the last bits of
    the fp32 accumulation order (conv3 walks K in steps of 16 here)."""
    B, H, W, gran = 4, 7, 256, 2
    cin = cout = 4 * W
    blk, hb = _block_pair(cin, W, gran, H)
    gm = seeded_bernoulli((B, W // gran), 0.62, 7)
    p = hb._prepare(torch.device(DEV))
    w2p, w3p = hb.tail_weights(p)
    _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, W // gran, gran, mask_in=gm.to(DEV))
    xn = F.relu(seeded_randn((B, H, H, cin), 3)).to(DEV)
    h1 = torch.empty(B, H, H, W, device=DEV)
    ops.bottleneck_head(xn, p["w1s"], idx, cnt, p["s1"], p["t1"], p["c1"], h1)
    ref = torch.empty(B, H, H, cout, device=DEV)
    ops.bottleneck_tail(h1, w2p, w3p, idx, cnt, p["s2"], p["t2_tab"], p["c2"], p["t3c"], ref, residual=xn)
    out = torch.empty_like(ref)
    ops.bottleneck_smallmap(xn, p["w1s"], w2p, w3p, idx, cnt, p["s1"], p["t1"], p["c1"], p["s2"], p["t2_tab"], p["c2"], p["t3c"], out,
                            residual=xn)
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-5), (out - ref).abs().max().item()


def test_smallmap_rejects_what_does_not_fit(ops):
    from laudnet_amd import LdnError
    assert not ops.bottleneck_smallmap_fits(9, 9, 2048, 512, 2048)      # 81 pixels
    assert not ops.bottleneck_smallmap_fits(8, 8, 2048, 512, 2048)      # 64 pixels at width 512: h1 / h2 leave no room for a weight slot
    assert not ops.bottleneck_smallmap_fits(7, 7, 2048, 1024, 4096)
    x = torch.zeros(1, 9, 9, 256, device=DEV)
    with pytest.raises(LdnError):
        z = torch.zeros(1, device=DEV)
        ops.bottleneck_smallmap(x, torch.zeros(1, dtype=torch.bfloat16, device=DEV), torch.zeros(1, dtype=torch.bfloat16, device=DEV),
                                torch.zeros(1, dtype=torch.bfloat16, device=DEV), torch.zeros(1, 64, dtype=torch.int32, device=DEV),
                                torch.zeros(1, dtype=torch.int32, device=DEV), z, z, z, z, z, z, z, torch.zeros(1, 9, 9, 256, device=DEV))
