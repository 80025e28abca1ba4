"""GPU parity of the fused bottleneck tail (ldn_bottleneck_tail: conv2 3x3 -> bn2/ReLU -> conv3 1x1 -> bn3 + residual + ReLU
in one launch, bf16x3 arithmetic) and of conv1's pre-split output format, against the dense-emulation algebra of the
reference (laud_resnet.py:115-144, channel mask applied before BN).  Tolerance 2e-4 + 1e-4 relative on O(1) activations
(north star: 1e-3)."""
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


def _decode_split(h1s):
    """[B,H,W,ld] fp32-typed storage of [octet][8 hi | 8 lo] bf16 -> fp32 values hi + lo, [B,H,W,ld]."""
    B, H, W, ld = h1s.shape
    raw = h1s.contiguous().view(torch.bfloat16).reshape(B, H, W, ld // 8, 2, 8).float()
    return (raw[..., 0, :] + raw[..., 1, :]).reshape(B, H, W, ld)


@pytest.mark.parametrize("B,H,cin,W,gran,st", [(4, 14, 1024, 256, 2, 1), (3, 28, 512, 128, 2, 1), (2, 56, 256, 64, 2, 1),
                                               (3, 56, 64, 64, 2, 1), (3, 14, 64, 256, 4, 1), (2, 9, 32, 64, 2, 1),
                                               (9, 14, 128, 256, 2, 1),
                                               # stride-2 3x3 (the first block of stages 2 / 3: laud_resnet.py:123 with stride 2), odd and
                                               # small maps, every width
                                               (3, 56, 256, 128, 2, 2), (3, 28, 512, 256, 2, 2), (2, 13, 64, 64, 2, 2),
                                               (2, 30, 64, 128, 4, 2), (2, 56, 32, 64, 2, 2), (5, 8, 64, 256, 2, 2)])
def test_tail_vs_reference_algebra(ops, B, H, cin, W, gran, st):
    ops.set_math_mode("bf16x3")      # the pre-split (bf16 hi | lo) formats are checked here; Bottleneck.tail_weights follows the mode
    try:
        _tail_vs_reference_algebra(ops, B, H, cin, W, gran, st)
    finally:
        ops.set_math_mode("fp32")


def _tail_vs_reference_algebra(ops, B, H, cin, W, gran, st):
    G = W // gran
    gm = seeded_bernoulli((B, G), 0.62, 31 + H)
    gm[0] = 0.0          # an image with no active channel
    gm[1] = 1.0          # an image with all channels
    cout = 4 * W
    Ho = (H - 1) // st + 1
    blk = TR.BottleneckRef(cin, W, stride=st, downsample=None, dyn_mode="channel", channel_dyn_granularity=gran,
                           channel_masker="MLP", output_size=Ho).eval()
    TR.randomize_bn_(blk, 5)
    with torch.no_grad():
        for m in (blk.conv1, blk.conv2, blk.conv3):
            m.weight.normal_(0, (2.0 / (m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3])) ** 0.5)
    x = F.relu(seeded_randn((B, cin, H, H), 32))
    ident = F.relu(seeded_randn((B, cout, Ho, Ho), 33))    # the residual (x itself when cin == cout, else a projection's output)
    cm = TR.broadcast_channel_mask(gm, W)
    with torch.no_grad():
        h1 = F.relu(blk.bn1(blk.conv1(x) * cm))
        h2 = F.relu(blk.bn2(blk.conv2(h1) * cm))
        want = F.relu(blk.bn3(blk.conv3(h2)) + ident).permute(0, 2, 3, 1)
    from laudnet_amd.laud_resnet import Bottleneck
    hb = Bottleneck(cin, W, stride=st, downsample=None, dyn_mode="channel", channel_dyn_granularity=gran,
                    channel_masker="MLP", output_size=Ho).eval()
    hb.load_state_dict(blk.state_dict())
    hb = hb.to(DEV)
    p = hb._prepare(torch.device(DEV))
    _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, G, gran, mask_in=gm.to(DEV))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    h1s = torch.full((B, H, H, W), float("nan"), device=DEV)
    ops.conv_image(xn, p["w1"], p["s1"], p["t1"], h1s, n_idx=idx, n_cnt=cnt, post_sub=p["c1"], relu=1, math="bf16x3", out_split=True)
    dec = _decode_split(h1s).cpu()
    c1 = p["c1"].cpu()
    for b in range(B):
        n = int(cnt[b])
        ch = idx[b, :n].cpu().long()
        want1 = h1[b, ch].permute(1, 2, 0) - c1[ch]
        assert torch.allclose(dec[b, :, :, :n], want1, atol=1e-4, rtol=1e-4), f"conv1 split output, image {b}"
        pad = (n + 31) // 32 * 32
        assert bool((dec[b, :, :, n:pad] == 0).all()), "columns up to the next multiple of 32 must be zero"
    w2p, w3p = hb.tail_weights(p)
    # conv1 on k_head: same pre-split output (identical arithmetic up to the order of the K sum)
    if cin % 32 == 0:
        h1h = torch.full((B, H, H, W), float("nan"), device=DEV)
        ops.bottleneck_head(xn, p["w1s"], idx, cnt, p["s1"], p["t1"], p["c1"], h1h)
        dech = _decode_split(h1h).cpu()
        for b in range(B):
            n = int(cnt[b])
            ch = idx[b, :n].cpu().long()
            want1 = h1[b, ch].permute(1, 2, 0) - c1[ch]
            assert torch.allclose(dech[b, :, :, :n], want1, atol=1e-4, rtol=1e-4), f"k_head output, image {b}"
            pad = (n + 31) // 32 * 32
            assert bool((dech[b, :, :, n:pad] == 0).all()), "k_head: columns up to the next multiple of 32 must be zero"
        h1s = h1h     # the tail below consumes k_head's output
    idn = ident.permute(0, 2, 3, 1).contiguous().to(DEV)
    splits = ops.bottleneck_tail_splits(H, H, W, st)
    assert splits > 0
    colsum = torch.full((B, splits, cout), float("nan"), device=DEV)
    out = torch.full((B, Ho, Ho, cout), float("nan"), device=DEV)
    ops.bottleneck_tail(h1s, w2p, w3p, idx, cnt, p["s2"], p["t2_tab"], p["c2"], p["t3c"], out, residual=idn, colsum=colsum, stride=st)
    torch.cuda.synchronize()
    err = (out.cpu() - want).abs()
    assert torch.allclose(out.cpu(), want, atol=2e-4, rtol=1e-4), f"max err {err.max().item():.3e} at {tuple(torch.nonzero(err == err.max())[0].tolist())}"
    assert torch.allclose(colsum.sum(dim=1).cpu().double(), out.cpu().double().sum(dim=(1, 2)), atol=1e-2, rtol=1e-5)
    # in-place residual stream (out aliases residual): same result
    ops.bottleneck_tail(h1s, w2p, w3p, idx, cnt, p["s2"], p["t2_tab"], p["c2"], p["t3c"], idn, residual=idn, stride=st)
    assert torch.equal(idn, out)


@pytest.mark.parametrize("B,H,Wd", [(3, 56, 56), (4, 12, 12), (2, 9, 20), (5, 28, 28)])
def test_folded_projection_block_vs_reference(ops, B, H, Wd):
    """Stage 1's first block (64 -> 64 -> 256, stride 1, projection shortcut): the shortcut folded into conv3's K loop
    (ldn_bottleneck_head_split + ldn_bottleneck_tail_proj) against the reference algebra (laud_resnet.py:115-144 with the downsample of
    :138-141) and against the unfolded execution (projection launch + residual read); x_split decodes to x."""
    import torch.nn as nn
    from laudnet_amd.laud_resnet import Bottleneck
    cin, W, gran = 64, 64, 2
    cout = 4 * W
    mk = lambda cls: cls(cin, W, stride=1, downsample=nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout)),
                         dyn_mode="channel", channel_dyn_granularity=gran, channel_masker="MLP", channel_masker_layers=2, output_size=H).eval()
    ref = mk(TR.BottleneckRef)
    TR.randomize_bn_(ref, 7)
    with torch.no_grad():
        for m in (ref.conv1, ref.conv2, ref.conv3, ref.downsample[0]):
            m.weight.normal_(0, (2.0 / (m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3])) ** 0.5)
    hb = mk(Bottleneck)
    hb.load_state_dict(ref.state_dict())
    hb = hb.to(DEV)
    gm = seeded_bernoulli((B, W // gran), 0.62, 41 + H)
    gm[0] = 0.0
    gm[1] = 1.0
    x = F.relu(seeded_randn((B, cin, H, Wd), 42))
    ref.forced_channel_mask = gm
    hb.forced_channel_mask = gm.to(DEV)
    st0 = lambda t: (t, None, None, None, None, None, torch.tensor(0.0, device=t.device))
    with torch.no_grad():
        want = ref(st0(x), 1.0)[0]
    assert ops.bottleneck_tail_proj_fits(H, Wd, W, cin)
    ops.set_math_mode("bf16x3")
    try:
        outs = {}
        for fold in (True, False):
            hb.use_folded_projection = fold
            seen = []
            orig = ops.bottleneck_tail_proj
            ops.bottleneck_tail_proj = lambda *a, **k: (seen.append(1), orig(*a, **k))[1]
            try:
                with torch.no_grad():
                    outs[fold] = hb(st0(x.to(DEV)), 1.0)[0].cpu()
            finally:
                ops.bottleneck_tail_proj = orig
            assert bool(seen) == fold, "the folded projection must run exactly when it is switched on"
        # x_split of the head launch decodes to x
        p = hb._prep
        _, idx, cnt, _ = ops.channel_masker(None, None, None, None, None, W // gran, gran, mask_in=gm.to(DEV))
        xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        h1 = torch.empty(B, H, Wd, W, device=DEV)
        xs = ops.x_split_buffer(B * H * Wd, cin, DEV)
        ops.bottleneck_head(xn, p["w1s"], idx, cnt, p["s1"], p["t1"], p["c1"], h1, x_split=xs)
        assert torch.allclose(ops.decode_x_split(xs, B * H * Wd, cin).cpu(), xn.reshape(-1, cin).cpu(), atol=0.0, rtol=2e-5)      # hi + lo = x to 2^-17
    finally:
        ops.set_math_mode("fp32")
    for fold in (True, False):
        err = (outs[fold] - want).abs().max().item()
        assert torch.allclose(outs[fold], want, atol=2e-4, rtol=1e-4), f"fold={fold}: max err {err:.3e}"
    assert torch.allclose(outs[True], outs[False], atol=1e-4, rtol=1e-4)
