"""Training under frozen BatchNorm statistics on the library's row kernels (laudnet_amd/training.py).

-m gpu: forward value and ALL gradients -- input, conv1 / conv2 / conv3 weights, BatchNorm affine parameters, the projection shortcut, and the
straight-through term of the hard mask -- of spatial, layer AND channel bottlenecks, stride 1 and stride 2 + projection, against the ORACLE's
autograd (oracle/torch_ref.py: the reference's dense emulation, BatchNorm in eval mode = frozen statistics) on the reference-generated block
fixtures of `blocks_s1.pt` / `blocks_s2.pt`, both arithmetic modes; whole models (`det_tiny.pt::channel_r50`, `::layer_r50` -- the shipped
detection configs -- and two classifiers of `full_tiny.pt`) in training mode with IDENTICAL Gumbel noise: outputs, statistics and the gradient of
every parameter (models/utils.py:56-58; lad_mmdet_resnet.py:753-758).  Tolerance: plain 1e-3 on values of scale <= 1, 1e-3 of the tensor's
scale above (the assert says which).  CPU: the transposed neighbour table against a brute-force adjoint, stride 1 and 2."""
import pytest
import torch

from fill import fill_state_dict, seeded_bernoulli, seeded_randn
from helpers import block_input, load_golden, make_block

DEV = "cuda:0"
BLOCKS = dict(load_golden("blocks_s1.pt"))
BLOCKS.update(load_golden("blocks_s2.pt"))
TRAIN_BLOCKS = ["spatial_g4_s1", "spatial_g1_s1", "layer_s1", "spatial_g1_s2", "spatial_g4_s2", "layer_s2",
                "channel_g1_s1", "channel_g2_s1", "channel_g1_s2", "channel_g2_s2"]


def _err(got, want):
    """max |got - want|, in units of max(1, max |want|): plain absolute error for O(1) tensors"""
    return (got - want).abs().max().item() / max(1.0, want.abs().max().item())


def _close(got, want, math_mode, what):
    """fp32 arithmetic: every element within 1e-3 (of max(1, scale)).  bf16x3 arithmetic (1e-5-class forward error): a pre-activation that sits
    within that error of zero takes the other side of its ReLU, and ONE flipped unit moves a 3x3 neighbourhood of d x (all channels) and one
    filter of the weight gradients by O(1) -- exactly what happens to the masker decisions of the inference path at near-ties.  There the
    bar is: at most 8 % of the elements outside the tolerance (one or two flipped units: a 16-filter 3x3 layer loses 6 % to one), relative Frobenius error below 5 %."""
    if math_mode != "bf16x3":
        assert _err(got, want) < 1e-3, f"{what}: {_err(got, want):.2e} (scale {want.abs().max().item():.2e})"
        return
    d = (got - want).abs()
    tol = 1e-3 * max(1.0, want.abs().max().item())
    frac = (d > tol).float().mean().item()
    fro = (d.norm() / want.norm().clamp(min=1e-12)).item()
    few = got.numel() < 2000      # (a small tensor -- a [B, G] mask gradient, a BatchNorm vector -- has no "3 % of the elements": its Frobenius error speaks)
    assert (few or frac <= 0.08) and fro < 0.05, f"{what}: {100 * frac:.2f} % of the elements outside 1e-3, relative Frobenius error {fro:.2e}"


def _start(x):
    return (x, None, None, None, None, None, torch.tensor(0.0, device=x.device))


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRAIN_BLOCKS)
def test_block_gradients_vs_oracle_autograd(name, math_mode):
    from laudnet_amd import ops
    from laudnet_amd.laud_resnet import Bottleneck
    from laudnet_amd.training import sparse_block_train
    from oracle import torch_ref as TR
    ops.set_math_mode(math_mode)
    try:
        fx = BLOCKS[name]
        channel = fx["kw"]["dyn_mode"] == "channel"
        hip = make_block(Bottleneck, fx).to(DEV)
        ref = make_block(TR.BottleneckRef, fx).to(DEV)          # eval mode: BatchNorm uses its running statistics (frozen)
        x0 = block_input(fx).to(DEV)
        mask0 = fx["channel_mask" if channel else "spatial_mask"].float().to(DEV)

        # oracle: autograd through the dense emulation with the same hard mask as a differentiable input
        xr = x0.clone().requires_grad_(True)
        mr = mask0.clone().requires_grad_(True)
        if channel:
            ref.forced_channel_mask = mr
        else:
            ref.forced_spatial_mask = mr
        for p_ in ref.parameters():
            p_.requires_grad_(True)
        out_r = ref(_start(xr), 1.0)[0]
        gout = seeded_randn(tuple(out_r.shape), 77).to(DEV)     # upstream gradient
        out_r.backward(gout)

        xh = x0.clone().requires_grad_(True)
        mh = mask0.clone().requires_grad_(True)
        for p_ in hip.parameters():
            p_.requires_grad_(True)
        out_h = sparse_block_train(hip, xh, mh)
        out_h.backward(gout)
        torch.cuda.synchronize()

        assert _err(out_h.detach(), out_r.detach()) < 1e-3, "forward"
        _close(xh.grad, xr.grad, math_mode, "d x")
        _close(mh.grad, mr.grad, math_mode, "straight-through term d mask")
        ref_grads = dict(ref.named_parameters())
        checked = 0
        for pname, ph in hip.named_parameters():
            if "masker" in pname:
                continue                                         # (the mask is an input here: the maskers are not part of the graph)
            want = ref_grads[pname].grad
            assert ph.grad is not None and want is not None, pname
            _close(ph.grad, want, math_mode, f"d {pname}")
            checked += 1
        assert checked >= 9 + (2 if fx["has_downsample"] else 0)   # three convs, three BatchNorms (weight + bias) [+ the projection]
        assert mr.grad.abs().max().item() > 0 and (mask0 < 0.5).any(), "the fixture must exercise dropped units"
    finally:
        ops.set_math_mode("fp32")


@pytest.mark.gpu
def test_training_scope_is_enforced():
    from laudnet_amd import LdnError
    from laudnet_amd.laud_resnet import Bottleneck
    from laudnet_amd.training import sparse_block_train
    fx = BLOCKS["both_s1"]
    blk = make_block(Bottleneck, fx).to(DEV)
    with pytest.raises(LdnError):                                # dyn_mode 'both' is not built
        sparse_block_train(blk, block_input(fx).to(DEV), fx["spatial_mask"].float().to(DEV))
    fx = BLOCKS["channel_g2_s1"]
    blk = make_block(Bottleneck, fx).to(DEV)
    with pytest.raises(LdnError):                                # a pixel mask for a channel block
        sparse_block_train(blk, block_input(fx).to(DEV), torch.ones(3, 1, 1, 1, device=DEV))
    blk.bn2.train()
    with pytest.raises(LdnError):                                # BatchNorm in batch-statistics mode
        sparse_block_train(blk, block_input(fx).to(DEV), fx["channel_mask"].float().to(DEV))
    with pytest.raises(LdnError):                                # no CPU path
        sparse_block_train(make_block(Bottleneck, fx), block_input(fx), fx["channel_mask"].float())


def _freeze_bn_train(model):
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    return model


def _compare_param_grads(hip, ref, math_mode="fp32"):
    want = dict(ref.named_parameters())
    n = 0
    for name, p_ in hip.named_parameters():
        w = want[name].grad
        if w is None and p_.grad is None:
            continue
        assert (w is None) == (p_.grad is None), f"{name}: gradient present on one side only"
        _close(p_.grad, w, math_mode, f"d {name}")
        n += 1
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["channel_r50", "layer_r50"])
def test_detection_backbone_train_step_vs_oracle(case):
    """The shipped detection configs (channel-2222 / layer): one training forward + backward of the whole backbone under norm_eval -- Gumbel
    masks sampled from its own maskers with the oracle's noise -- against the oracle's autograd: stage outputs, statistics, FLOPs, and the
    gradient of EVERY parameter (convolutions, BatchNorm affine terms, projections, maskers) of a loss that touches the four feature maps and
    the FLOPs ratio (single_stage.py:44-90 adds (flops / dense_flops - target)^2)."""
    from laudnet_amd import ops
    from laudnet_amd.detection import LAD_MMDet_ResNet
    from laudnet_amd.training import prepare_for_training, train_forward
    from oracle import det_ref as DR
    ops.set_math_mode("fp32")     # (whole models in true-fp32 arithmetic: the Gumbel samples and every ReLU decision must coincide with the oracle's)
    try:
        fx = load_golden("det_tiny.pt")[case]
        ref = DR.LADDetResNetRef(**fx["kw"])
        ref.load_state_dict(fill_state_dict(ref.state_dict(), fx["seed"]))
        hip = LAD_MMDet_ResNet(**fx["kw"])
        hip.load_state_dict(fill_state_dict(hip.state_dict(), fx["seed"]))
        ref, hip = _freeze_bn_train(ref.to(DEV)), prepare_for_training(hip.to(DEV))
        x = seeded_randn(tuple(fx["shape"]), fx["x_seed"]).to(DEV)
        gs = None

        def loss_of(res):
            nonlocal gs
            outs, add, _ = res
            if gs is None:
                gs = [seeded_randn(tuple(o.shape), 500 + i).to(DEV) for i, o in enumerate(outs)]
            return sum((o * g).sum() for o, g in zip(outs, gs)) / 100.0 + 10.0 * (add["flops"] / add["dense_flops"] - 0.5) ** 2

        torch.manual_seed(1234)
        res_r = ref(x)
        loss_of(res_r).backward()
        torch.manual_seed(1234)
        res_h = train_forward(hip, x)
        loss_of(res_h).backward()
        torch.cuda.synchronize()
        for a, b in zip(res_h[0], res_r[0]):
            assert _err(a.detach(), b.detach()) < 1e-3, "stage output"
        for k in ("spatial_sparsity_conv3", "spatial_sparsity_conv2", "spatial_sparsity_conv1", "channel_sparsity"):
            for a, b in zip(res_h[1][k], res_r[1][k]):
                assert torch.allclose(a.detach().float(), b.detach().float(), atol=1e-6), k     # identical Gumbel samples
        assert torch.allclose(res_h[1]["flops_perc_list"].detach(), res_r[1]["flops_perc_list"].detach(), atol=1e-5)
        assert abs(float(res_h[1]["flops"].detach()) - float(res_r[1]["flops"].detach())) <= 1e-5 * float(res_r[1]["flops"].detach())
        assert abs(float(res_h[1]["dense_flops"]) - float(res_r[1]["dense_flops"])) <= 1e-6 * float(res_r[1]["dense_flops"])
        n = _compare_param_grads(hip, ref)
        assert n >= 100, n
        drops = sum(float((1 - v).sum()) for k in ("spatial_sparsity_conv3", "channel_sparsity") for v in res_r[1][k])
        assert drops > 0, "the sampled masks must drop something"
    finally:
        ops.set_math_mode("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["r101_channel2222", "r101_layer", "r101_spatial4421"])
def test_classifier_train_step_vs_oracle(case):
    """The classifier's training forward (7-tuple) + backward under frozen BatchNorm statistics, the sparsity criterion on the tuple."""
    from laudnet_amd import ops, sparsity_loss
    import laudnet_amd
    from laudnet_amd.training import prepare_for_training, train_forward
    from oracle import torch_ref as TR
    ops.set_math_mode("fp32")
    fx = load_golden("full_tiny.pt")[case]
    depth = 101 if "101" in fx["factory"] else 50
    ref = (TR.resnet101_ref if depth == 101 else TR.resnet50_ref)(**fx["kw"])
    hip = (laudnet_amd.uni_resnet101 if depth == 101 else laudnet_amd.uni_resnet50)(**fx["kw"])
    sd = fill_state_dict(ref.state_dict(), fx["seed"])
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.3                                   # damped residual branches: O(1) activations through 33 blocks
    ref.load_state_dict(sd)
    hip.load_state_dict(sd)
    ref, hip = _freeze_bn_train(ref.to(DEV)), prepare_for_training(hip.to(DEV))
    x = seeded_randn((fx["batch"], 3, 224, 224), fx["x_seed"]).to(DEV)
    g = seeded_randn((fx["batch"], fx["kw"].get("num_classes", 1000)), 9).to(DEV)

    def loss_of(out):
        return (out[0] * g).sum() / 10.0 + 10.0 * (out[5].mean() - 0.5) ** 2 + 1e-18 * out[6] ** 2

    torch.manual_seed(77)
    out_r = ref(x, 1.0)
    loss_of(out_r).backward()
    torch.manual_seed(77)
    out_h = train_forward(hip, x, 1.0)
    loss_of(out_h).backward()
    torch.cuda.synchronize()
    assert _err(out_h[0].detach(), out_r[0].detach()) < 1e-3, "logits"
    for i in (1, 2, 3, 4):
        for a, b in zip(out_h[i], out_r[i]):
            assert torch.allclose(a.detach().float(), b.detach().float(), atol=1e-6), i
    assert torch.allclose(out_h[5].detach(), out_r[5].detach(), atol=1e-5)
    assert abs(float(out_h[6]) - float(out_r[6])) <= 1e-5 * float(out_r[6])
    n = _compare_param_grads(hip, ref)
    assert n >= 200, n


def test_transposed_neighbour_table_is_the_adjoint_cpu():
    """nbrT is the adjoint of nbr: for random g and h,  sum_m sum_t g[m] h[nbr[m, t]] w[t]  ==  sum_r h[r] sum_t g[nbrT[r, t]] w[t]."""
    from types import SimpleNamespace
    from laudnet_amd.training import transposed_neighbour_table
    from oracle import index_ref as IR
    import numpy as np
    B, H, W, S = 3, 6, 5, 3
    patch = seeded_bernoulli((B, S, S), 0.5, 5).numpy()
    m3 = IR.upsample_patch_mask(patch, H, W).astype(bool)
    m1 = IR.dilate_mask(m3, 1, 1)
    idx3, _ = IR.nonzero_rows(m3)
    idx1, _ = IR.nonzero_rows(m1)
    pos3 = IR.position_map(m3).reshape(-1)
    nbr = IR.neighbour_table(m3, m1, 1)
    n3, n1 = len(idx3), len(idx1)
    ix = SimpleNamespace(idx1=torch.from_numpy(idx1.astype(np.int32)), pos3=torch.from_numpy(pos3.astype(np.int32)), cap1=n1,
                         cnt=torch.tensor([n3, n1], dtype=torch.int32))
    nbrT = transposed_neighbour_table(ix, B, H, W).numpy()
    g, h, w = np.random.RandomState(0).randn(n3), np.random.RandomState(1).randn(n1), np.random.RandomState(2).randn(9)
    nb = nbr.reshape(n3, 9)
    lhs = sum(g[m] * h[nb[m, t]] * w[t] for m in range(n3) for t in range(9) if nb[m, t] >= 0)
    rhs = sum(h[r] * g[nbrT[r, t]] * w[t] for r in range(n1) for t in range(9) if nbrT[r, t] >= 0)
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_transposed_neighbour_table_stride2_cpu():
    """the same adjoint identity for a stride-2 3x3 (input 8 x 6 -> output 4 x 3)"""
    from types import SimpleNamespace
    from laudnet_amd.training import transposed_neighbour_table
    from oracle import index_ref as IR
    import numpy as np
    B, Ho, Wo, st = 2, 4, 3, 2
    Hi, Wi = Ho * st, Wo * st
    patch = seeded_bernoulli((B, 2, 1), 0.6, 11).numpy()
    m3 = IR.upsample_patch_mask(patch, Ho, Wo).astype(bool)
    m1 = IR.dilate_mask(m3, st, 1)
    idx3, _ = IR.nonzero_rows(m3)
    idx1, _ = IR.nonzero_rows(m1)
    pos3 = IR.position_map(m3).reshape(-1)
    nbr = IR.neighbour_table(m3, m1, st)
    n3, n1 = len(idx3), len(idx1)
    ix = SimpleNamespace(idx1=torch.from_numpy(idx1.astype(np.int32)), pos3=torch.from_numpy(pos3.astype(np.int32)), cap1=n1,
                         cnt=torch.tensor([n3, n1], dtype=torch.int32))
    nbrT = transposed_neighbour_table(ix, B, Hi, Wi, st, Ho, Wo).numpy()
    g, h, w = np.random.RandomState(0).randn(n3), np.random.RandomState(1).randn(n1), np.random.RandomState(2).randn(9)
    nb = nbr.reshape(n3, 9)
    lhs = sum(g[m] * h[nb[m, t]] * w[t] for m in range(n3) for t in range(9) if nb[m, t] >= 0)
    rhs = sum(h[r] * g[nbrT[r, t]] * w[t] for r in range(n1) for t in range(9) if nbrT[r, t] >= 0)
    assert n3 > 0 and abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))
