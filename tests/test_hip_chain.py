"""GPU tests of ldn_bottleneck_chain (k_chain): a run of stride-1 channel-mode bottlenecks (stage 3: 14x14 maps) executed as ONE
launch, workgroup b walking image b through masker -> conv1 -> conv2/conv3 of every block (laud_resnet.py:104-147 block after
block).  The chain runs the SAME device code as ldn_channel_masker / ldn_bottleneck_head / ldn_bottleneck_tail, so the bar is
bit-identity with the block-by-block execution: logits, every block's mask and channel list, the statistics.  (Parity of the
chained model with the oracle is the headline test of tests/test_hip_fullsize.py, which runs with the chain on.)"""
import os
import sys

import pytest
import torch

from fill import fill_state_dict, seeded_randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(factory, width_mult, batch, seed=3):
    import laudnet_amd
    from laudnet_amd import ops
    sys.path.insert(0, ROOT)
    import bench
    ops.set_math_mode("bf16x3")
    kw = dict(dyn_mode=["channel"] * 4, channel_dyn_granularity=[2, 2, 2, 2], channel_masker=["MLP"] * 4,
              channel_masker_layers=[2, 2, 2, 2], reduction_ratio=[16] * 4, num_classes=1000, input_size=224, width_mult=width_mult)
    m = getattr(laudnet_amd, factory)(**kw).eval()
    sd = fill_state_dict(m.state_dict(), seed)
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.3
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = seeded_randn((batch, 3, 224, 224), 77).to(DEV).contiguous(memory_format=torch.channels_last)
    bench.calibrate_maskers(m, x, 0.62, None)
    return m, x


@pytest.fixture(autouse=True)
def _restore():
    from laudnet_amd import ops
    from laudnet_amd.laud_resnet import Bottleneck, ResNet
    widths = Bottleneck.fused_head_widths          # the library default (LDN_HEAD_WIDTHS), whatever it is
    yield
    ops.set_math_mode("fp32")
    ResNet.use_chain = True
    Bottleneck.fused_head_widths = widths


def _blocks(m):
    return [b for s in (1, 2, 3, 4) for b in getattr(m, f"layer{s}")]


def _run(m, x, chain):
    from laudnet_amd.laud_resnet import ResNet
    ResNet.use_chain = chain
    with torch.no_grad():
        out = m(x, 1.0)
    torch.cuda.synchronize()
    masks = [b.last_channel_mask.clone() for b in _blocks(m)]
    cnts = [b.last_channel_cnt.clone() for b in _blocks(m)]
    return out, masks, cnts


def _same(a, b):
    assert torch.equal(a[0][0], b[0][0]), f"logits differ: max {float((a[0][0] - b[0][0]).abs().max()):.3e}"
    for ga, gb in zip(a[0][1:5], b[0][1:5]):
        for ta, tb in zip(ga, gb):
            assert torch.equal(ta, tb)
    assert torch.equal(a[0][5], b[0][5]) and torch.equal(a[0][6], b[0][6])
    for i, (ma, mb) in enumerate(zip(a[1], b[1])):
        assert torch.equal(ma, mb), f"block {i}: masks differ"
    for i, (ca, cb) in enumerate(zip(a[2], b[2])):
        assert torch.equal(ca, cb), f"block {i}: channel counts differ"


@pytest.mark.parametrize("factory,width_mult,batch", [("uni_resnet101", 1.0, 24), ("uni_resnet50", 0.5, 9), ("uni_resnet50", 0.25, 16)])
def test_chain_equals_block_by_block(factory, width_mult, batch):
    """Stage-3 widths 256 / 128 / 64 (k_chain<8> / <4> / <2>), batches that are and are not multiples of the XCD count."""
    from laudnet_amd.laud_resnet import Bottleneck
    Bottleneck.fused_head_widths = (64, 128, 256)      # the block-by-block run uses k_head wherever the chain does
    m, x = _model(factory, width_mult, batch)
    calls = []
    from laudnet_amd import ops
    orig = ops.bottleneck_chain
    ops.bottleneck_chain = lambda *a, **k: (calls.append(a[2].shape[0]), orig(*a, **k))[1]
    try:
        chained = _run(m, x, True)
    finally:
        ops.bottleneck_chain = orig
    n3 = len(m.layer3)
    assert calls == [n3 - 1], f"stage 3 must run as one chain of {n3 - 1} blocks (got {calls})"
    plain = _run(m, x, False)
    _same(chained, plain)
    keep = float(torch.stack([mk.mean() for mk in chained[1]]).mean())
    assert 0.45 < keep < 0.8, "the maskers must be making real decisions"
    again = _run(m, x, True)        # run to run: bit-identical
    _same(chained, again)


def test_chain_without_inplace_residual_and_with_a_forced_mask_in_the_middle():
    """inplace_residual = False: the run's input stays intact (x_work is a separate tensor).  A forced mask on a block splits the
    run: the blocks before it chain, it runs on its own, the rest chain again."""
    from laudnet_amd import ops
    m, x = _model("uni_resnet101", 1.0, 8)
    m.inplace_residual = False
    base = _run(m, x, False)
    taps = []
    orig = ops.bottleneck_chain

    def spy(x_in, x_work, *a, **k):
        keep = x_in.clone()
        r = orig(x_in, x_work, *a, **k)
        torch.cuda.synchronize()
        taps.append((x_in.data_ptr() != x_work.data_ptr(), torch.equal(keep, x_in)))
        return r

    ops.bottleneck_chain = spy
    try:
        chained = _run(m, x, True)
    finally:
        ops.bottleneck_chain = orig
    assert taps == [(True, True)]
    _same(chained, base)
    m.inplace_residual = True
    blk = m.layer3[9]
    blk.forced_channel_mask = base[1][len(m.layer1) + len(m.layer2) + 9].clone()     # its own decision, but forced
    calls = []
    ops.bottleneck_chain = lambda *a, **k: (calls.append(a[2].shape[0]), orig(*a, **k))[1]
    try:
        split = _run(m, x, True)
    finally:
        ops.bottleneck_chain = orig
        blk.forced_channel_mask = None
    assert calls == [8, len(m.layer3) - 10]
    _same(split, base)


def test_chain_rejects_bad_arguments():
    from laudnet_amd import ops
    from laudnet_amd._lib import LdnError
    x = torch.zeros(2, 20, 20, 64, device=DEV)             # 400 pixels: does not fit one workgroup
    table = torch.zeros(2, ops.CHAIN_BLOCK_FIELDS, dtype=torch.int64, device=DEV)
    gap = torch.zeros(2, 8, 64, device=DEV)
    with pytest.raises(LdnError):
        ops.bottleneck_chain(x, x, table, 64, 16, 32, 2, gap)
    with pytest.raises(LdnError):
        ops.bottleneck_chain(x[:, :14, :14].contiguous(), x[:, :14, :14].contiguous(), table[:, :5].contiguous(), 64, 16, 32, 2, gap)
