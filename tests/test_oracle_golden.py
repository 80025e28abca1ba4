"""Pin the oracle (oracle/torch_ref.py, oracle/index_ref.py) to the golden fixtures that
were produced by importing the reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import (assert_tuple_close, block_input, full_model_blocks, injected_masks_for, load_golden,
                     make_block, start_state)
from fill import fill_state_dict, seeded_randn
from oracle import index_ref as IR
from oracle import torch_ref as TR

L1 = load_golden("l1_ops.pt")
BLOCKS = {**load_golden("blocks_s1.pt"), **load_golden("blocks_s2.pt"), **load_golden("blocks_extra.pt")}
FULL = load_golden("full_tiny.pt")


# ------------------------------------------------------------------ L1 ops
def test_channel_mask_broadcast():
    for key in ("chan_mask", "chan_mask_full"):
        fx = L1[key]
        y = fx["x"] * TR.broadcast_channel_mask(fx["mask"], fx["x"].shape[1])
        assert torch.equal(y, fx["y"])
    # probed fact: [1,0,1] on 6 channels -> 1,1,0,0,1,1
    m = TR.broadcast_channel_mask(torch.tensor([[1., 0., 1.]]), 6).flatten()
    assert m.tolist() == [1, 1, 0, 0, 1, 1]


def test_spatial_mask_broadcast():
    for g in (1, 2, 6):
        fx = L1[f"spat_mask_g{g}"]
        y = fx["x"] * TR.broadcast_spatial_mask(fx["mask"], fx["x"].shape[1])
        assert torch.equal(y, fx["y"])


def test_expand_mask_torch_and_numpy():
    for fx in L1["expand"]:
        y = TR.expand_mask(fx["mask"], fx["stride"], fx["padding"])
        assert y.dtype == torch.bool and torch.equal(y, fx["y"]), (fx["stride"], fx["padding"], fx["groups"])
        if fx["groups"] == 1:
            ynp = IR.dilate_mask(fx["mask"][:, 0].numpy() > 0.5, fx["stride"], fx["padding"])
            assert np.array_equal(ynp, fx["y"][:, 0].numpy())


def test_nearest_index():
    for fx in L1["nearest"]:
        s, h = fx["s"], fx["h"]
        idx = IR.nearest_src_index(h, s)
        want = fx["y"][0, 0].numpy()
        got = idx[:, None] * s + idx[None, :]
        assert np.array_equal(got, want), (s, h)


def test_adaptive_pool_bins():
    for fx in L1["adaptive_pool"]:
        h, s = fx["h"], fx["s"]
        st, en = IR.adaptive_pool_bins(h, s)
        x = fx["x"].numpy()
        got = np.zeros((1, 2, s, s), dtype=np.float64)
        for i in range(s):
            for j in range(s):
                got[:, :, i, j] = x[:, :, st[i]:en[i], st[j]:en[j]].mean(axis=(2, 3), dtype=np.float64)
        assert np.allclose(got, fx["y"].numpy(), atol=1e-6)


def test_maskers():
    for fx in L1["maskers"]:
        if fx["kind"] == "spatial":
            m = TR.SpatialMaskerRef(fx["cin"], fx["groups"], fx["mask_size"]).eval()
        elif fx["kind"] == "mlp":
            m = TR.ChannelMaskerMLPRef(fx["cin"], fx["groups"], fx["layers"], fx["reduction"]).eval()
        else:
            m = TR.ChannelMaskerConvLinearRef(fx["cin"], fx["groups"], fx["reduction"]).eval()
        m.load_state_dict(fx["sd"], strict=True)
        with torch.no_grad():
            mask, sp, fl = m(fx["x"], 1.0)
        assert torch.equal(mask, fx["mask"]), fx["kind"]
        assert torch.equal(sp, fx["sparsity"])
        assert int(fl) == fx["flops"]
        if "logits" in fx and fx["kind"] != "conv_linear":
            with torch.no_grad():
                lg = m.logits(fx["x"])
            assert torch.allclose(lg.reshape(fx["logits"].shape), fx["logits"], atol=1e-6)


def test_tie_keeps():
    fx = [f for f in L1["maskers"] if f["kind"] == "spatial" and f["cin"] == 4][0]
    assert float(fx["mask"].min()) == 1.0  # reference: logit_keep >= logit_drop keeps on ties


# ------------------------------------------------------------------ L2 blocks
@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_block_masker_run(name):
    fx = BLOCKS[name]
    blk = make_block(TR.BottleneckRef, fx)
    with torch.no_grad():
        got = blk(start_state(block_input(fx)), 1.0)
    assert_tuple_close(got, fx["masker_run"], atol=1e-5, what=name)


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_block_injected_run(name):
    fx = BLOCKS[name]
    blk = make_block(TR.BottleneckRef, fx)
    blk.forced_spatial_mask = fx.get("spatial_mask")
    blk.forced_channel_mask = fx.get("channel_mask")
    with torch.no_grad():
        got = blk(start_state(block_input(fx)), 1.0)
    assert_tuple_close(got, fx["injected_run"], atol=1e-5, what=name)


@pytest.mark.parametrize("name", sorted(n for n in BLOCKS if "idx3" in BLOCKS[n]))
def test_block_index_lists(name):
    """The packed gather lists the HIP path must reproduce, vs torch.nonzero of the reference's masks."""
    fx = BLOCKS[name]
    kw = fx["kw"]
    patch = fx["spatial_mask"][:, 0].numpy() > 0.5
    m3 = IR.upsample_patch_mask(patch, kw["output_size"])
    assert np.array_equal(m3, fx["mask3_px"][:, 0].numpy())
    m1 = IR.dilate_mask(m3, kw["stride"], 1)
    assert np.array_equal(m1, fx["mask1_px"][:, 0].numpy())
    idx3, pre3 = IR.nonzero_rows(m3)
    idx1, pre1 = IR.nonzero_rows(m1)
    assert np.array_equal(idx3, fx["idx3"].numpy()) and np.array_equal(idx1, fx["idx1"].numpy())
    assert pre3[-1] == len(idx3) and pre1[-1] == len(idx1)
    # every in-bounds tap of an active output pixel is present in the dilated list
    nbr = IR.neighbour_table(m3, m1, kw["stride"])
    b, oy, ox = np.nonzero(m3)
    hin = kw["output_size"] * kw["stride"]
    for t in range(9):
        iy, ix = oy * kw["stride"] - 1 + t // 3, ox * kw["stride"] - 1 + t % 3
        inb = (iy >= 0) & (iy < hin) & (ix >= 0) & (ix < hin)
        assert np.all((nbr[:, t] >= 0) == inb)
        flat = (b * hin + iy) * hin + ix
        assert np.array_equal(idx1[nbr[inb, t]], flat[inb])


# ------------------------------------------------------------------ L3 full tiny models
def _build_full(fx, damp=None):
    layers = [3, 4, 6, 3] if fx["factory"] == "uni_resnet50" else [3, 4, 23, 3]
    model = TR.ResNetRef(layers, **fx["kw"]).eval()
    assert list(model.state_dict().keys()) == fx["keys"], "state_dict keys must equal the reference's"
    assert "n_params" not in fx or sum(p.numel() for p in model.parameters()) == fx["n_params"]
    sd = fill_state_dict(model.state_dict(), fx["seed"])
    if damp is not None:
        from fill import damp_residual_branches
        sd = damp_residual_branches(sd, damp)
    model.load_state_dict(sd)
    x = seeded_randn((fx["batch"], 3, fx["kw"]["input_size"], fx["kw"]["input_size"]), fx["x_seed"])
    return model, x


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_models_masker(name):
    fx = FULL[name]
    model, x = _build_full(fx)
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["masker_run"], atol=2e-4, rtol=1e-5, what=name)


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_models_injected(name):
    fx = FULL[name]
    model, x = _build_full(fx)
    blocks = full_model_blocks(model)
    masks = injected_masks_for(blocks, fx["batch"], fx["mask_seed"])
    for bname, blk in blocks:
        blk.forced_spatial_mask = masks[bname].get("spatial")
        blk.forced_channel_mask = masks[bname].get("channel")
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["injected_run"], atol=2e-4, rtol=1e-5, what=name)


# damped twins (tests/golden/make_full_damped_golden.py: bn3.weight * 0.3, logits O(1)): PLAIN absolute tolerances
FULL_DAMPED = load_golden("full_tiny_damped.pt")


@pytest.mark.parametrize("name", sorted(FULL_DAMPED))
def test_full_models_damped(name):
    fx = FULL_DAMPED[name]
    model, x = _build_full(fx, damp=fx["damp"])
    assert float(fx["injected_run"][0].abs().max()) < 5.0          # the recipe keeps the logits O(1)
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["masker_run"], atol=1e-5, rtol=1e-5, what=name + " masker run")
    blocks = full_model_blocks(model)
    masks = injected_masks_for(blocks, fx["batch"], fx["mask_seed"])
    for bname, blk in blocks:
        blk.forced_spatial_mask = masks[bname].get("spatial")
        blk.forced_channel_mask = masks[bname].get("channel")
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["injected_run"], atol=1e-5, rtol=1e-5, what=name + " injected run")


# ------------------------------------------------------------------ LAD-RegNet (torchvision containers restated)
REGNET = load_golden("regnet_tiny.pt")


def test_regnet_block_params_match_reference():
    from oracle import regnet_ref as RR
    for name, want in REGNET["params"].items():
        got = RR.regnet_block_params(**RR.REGNET_Y[name])
        assert got == want, name
    p800 = RR.regnet_block_params(**RR.REGNET_Y["lad_regnet_y_800mf"])
    assert p800["widths"] == [64, 144, 320, 784] and p800["depths"] == [1, 3, 8, 2]


def _build_regnet(fx):
    from oracle import regnet_ref as RR
    model = RR.RegNetRef(REGNET["tiny_params"] | {}, se_ratio=REGNET["tiny_params"]["se_ratio"], **fx["kw"]).eval()
    assert list(model.state_dict().keys()) == fx["keys"], "state_dict keys must equal the reference's"
    assert sum(p.numel() for p in model.parameters()) == fx["n_params"]
    model.load_state_dict(fill_state_dict(model.state_dict(), fx["seed"]))
    size = fx["kw"]["input_size"]
    return model, seeded_randn((fx["batch"], 3, size, size), fx["x_seed"])


@pytest.mark.parametrize("name", sorted(REGNET["cases"]))
def test_regnet_masker_run(name):
    fx = REGNET["cases"][name]
    model, x = _build_regnet(fx)
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["masker_run"], atol=2e-5, rtol=1e-5, what=name)


@pytest.mark.parametrize("name", sorted(REGNET["cases"]))
def test_regnet_injected_run(name):
    fx = REGNET["cases"][name]
    model, x = _build_regnet(fx)
    blocks = [(n, b.f) for n, b in model.blocks()]
    masks = injected_masks_for(blocks, fx["batch"], fx["mask_seed"])
    for bname, f in blocks:
        f.forced_spatial_mask = masks[bname].get("spatial")
        f.forced_channel_mask = masks[bname].get("channel")
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["injected_run"], atol=2e-5, rtol=1e-5, what=name)


# ---- LAD-RegNet with two spatial mask groups per block (reference-generated regnet_extra.pt, make_regnet_groups_golden.py)
REGNET_X = load_golden("regnet_extra.pt")


@pytest.mark.parametrize("name", sorted(REGNET_X["cases"]))
def test_regnet_two_mask_groups(name):
    fx = REGNET_X["cases"][name]
    assert fx["kw"]["spatial_mask_channel_group"] == [2, 2, 2, 2]
    model, x = _build_regnet(fx)
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["masker_run"], atol=2e-5, rtol=1e-5, what=name + " masker run")
    blocks = [(n, b.f) for n, b in model.blocks()]
    masks = injected_masks_for(blocks, fx["batch"], fx["mask_seed"])
    for bname, f in blocks:
        f.forced_spatial_mask = masks[bname].get("spatial")
        f.forced_channel_mask = masks[bname].get("channel")
    with torch.no_grad():
        got = model(x, 1.0)
    assert_tuple_close(got, fx["injected_run"], atol=2e-5, rtol=1e-5, what=name + " injected run")

