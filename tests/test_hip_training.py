"""Training on the packed kernels under frozen BatchNorm (laudnet_amd/training.py; SURVEY 8f-4 as scoped by VERDICT round 4, item 9).

-m gpu: forward value and ALL gradients (input, conv1 / conv2 / conv3 weights, and the straight-through term of the hard mask) of a spatial
and a layer-skip bottleneck against the ORACLE's autograd (oracle/torch_ref.py: the reference's dense emulation, BatchNorm in eval mode = frozen
statistics) on the reference-generated block fixtures `blocks_s1.pt::spatial_g4_s1`, `::spatial_g1_s1` and `::layer_s1`, both arithmetic
modes, 1e-3 (models/utils.py:56-58; lad_mmdet_resnet.py:753-758).  CPU: the transposed neighbour table against a brute-force adjoint."""
import pytest
import torch

from fill import seeded_bernoulli, seeded_randn
from helpers import block_input, load_golden, make_block

DEV = "cuda:0"
BLOCKS = load_golden("blocks_s1.pt")


def _rel_err(got, want):
    return (got - want).abs().max().item() / max(1.0, want.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["spatial_g4_s1", "spatial_g1_s1", "layer_s1"])
def test_sparse_block_gradients_vs_oracle_autograd(name, math_mode):
    from laudnet_amd import ops
    from laudnet_amd.laud_resnet import Bottleneck
    from laudnet_amd.training import sparse_block_train
    from oracle import torch_ref as TR
    ops.set_math_mode(math_mode)
    try:
        fx = BLOCKS[name]
        hip = make_block(Bottleneck, fx).to(DEV)
        ref = make_block(TR.BottleneckRef, fx).to(DEV)          # eval mode: BatchNorm uses its running statistics (frozen)
        x0 = block_input(fx).to(DEV)
        gout = seeded_randn(tuple(x0.shape), 77).to(DEV)        # upstream gradient (fixed "Gumbel noise": the mask is the fixture's hard sample)
        mask0 = fx["spatial_mask"].float().to(DEV)

        # oracle: autograd through the dense emulation with the same hard mask as a differentiable input
        xr = x0.clone().requires_grad_(True)
        mr = mask0.clone().requires_grad_(True)
        ref.forced_spatial_mask = mr
        for p_ in ref.parameters():
            p_.requires_grad_(True)
        out_r = ref((xr, None, None, None, None, None, torch.tensor(0.0, device=DEV)), 1.0)[0]
        out_r.backward(gout)

        xh = x0.clone().requires_grad_(True)
        mh = mask0.clone().requires_grad_(True)
        for p_ in hip.parameters():
            p_.requires_grad_(True)
        out_h = sparse_block_train(hip, xh, mh)
        out_h.backward(gout)
        torch.cuda.synchronize()

        assert _rel_err(out_h.detach(), out_r.detach()) < 1e-3, "forward"
        assert _rel_err(xh.grad, xr.grad) < 1e-3, "d x"
        for conv in ("conv1", "conv2", "conv3"):
            got, want = getattr(hip, conv).weight.grad, getattr(ref, conv).weight.grad
            assert got is not None and _rel_err(got, want) < 1e-3, f"d {conv}.weight"
        assert _rel_err(mh.grad, mr.grad) < 1e-3, "straight-through term d mask"
        assert mr.grad.abs().max().item() > 0 and (mask0 < 0.5).any(), "the fixture must exercise dropped units"
    finally:
        ops.set_math_mode("fp32")


@pytest.mark.gpu
def test_sparse_block_train_scope_is_enforced():
    from laudnet_amd import LdnError
    from laudnet_amd.laud_resnet import Bottleneck
    from laudnet_amd.training import sparse_block_train
    fx = BLOCKS["channel_g2_s1"]
    blk = make_block(Bottleneck, fx).to(DEV)
    with pytest.raises(LdnError):
        sparse_block_train(blk, block_input(fx).to(DEV), torch.ones(3, 1, 1, 1, device=DEV))
    fx2 = load_golden("blocks_s2.pt")
    name = next(n for n in sorted(fx2) if fx2[n]["kw"]["dyn_mode"] == "spatial")
    blk2 = make_block(Bottleneck, fx2[name]).to(DEV)
    with pytest.raises(LdnError):
        sparse_block_train(blk2, block_input(fx2[name]).to(DEV), fx2[name]["spatial_mask"].float().to(DEV))


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
