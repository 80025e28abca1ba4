"""Token skipping (BASELINE config 5) on the HIP path against oracle/adavit_ref.py -- a SELF-CONSISTENCY check: the reference has no
model code for this configuration (parity unpinned, see the oracle's header).  Packed attention alone, one block, and a DeiT-S
shaped trunk; ragged token counts, an image that keeps only its CLS token, an image that keeps everything."""
import pytest
import torch

from fill import seeded_bernoulli, seeded_randn
from oracle import adavit_ref as AR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _keep(B, L, p, seed):
    k = seeded_bernoulli((B, L), p, seed)
    k[:, 0] = 1.0            # CLS
    if B > 1:
        k[1, 1:] = 0.0       # an image that keeps only CLS
    if B > 2:
        k[2] = 1.0           # an image that keeps everything
    return k


@pytest.mark.parametrize("B,L,heads,p", [(4, 197, 6, 0.5), (3, 64, 2, 0.3), (2, 256, 1, 0.7), (5, 50, 3, 0.5)])
def test_packed_mha_vs_dense_masked_attention(B, L, heads, p):
    from laudnet_amd import ops, load_library
    load_library()
    dim = 64 * heads
    qkv = seeded_randn((B, L, 3 * dim), 3 + L)
    keep = _keep(B, L, p, 5 + L)
    q, k, v = qkv.reshape(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 64 ** -0.5
    s = s.masked_fill(keep[:, None, None, :] < 0.5, float("-inf"))
    want = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, L, dim)
    tok_rows, prefix, count = ops.token_lists(keep.to(DEV))
    got = ops.packed_mha(qkv.reshape(B * L, 3 * dim).to(DEV), tok_rows, prefix, B, heads, L)
    n = int(count.item())
    assert n == int(keep.sum().item())
    rows = tok_rows[:n].long().cpu()
    assert torch.equal(rows, torch.nonzero(keep.reshape(-1)).reshape(-1))
    err = (got[:n].cpu() - want.reshape(B * L, dim)[rows]).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("B,L,dim,heads", [(4, 197, 384, 6), (3, 40, 128, 2)])
def test_token_skip_block_vs_oracle(B, L, dim, heads):
    from laudnet_amd import ops
    from laudnet_amd.adavit import TokenSkipBlock
    ref = AR.TokenSkipBlockRef(dim, heads).eval()
    torch.manual_seed(7)
    for p_ in ref.parameters():
        torch.nn.init.normal_(p_, std=0.05) if p_.dim() > 1 else torch.nn.init.normal_(p_, mean=1.0 if "norm" in "" else 0.0, std=0.1)
    hip = TokenSkipBlock(dim, heads).eval()
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV)
    x = seeded_randn((B, L, dim), 21)
    keep = _keep(B, L, 0.5, 22)
    with torch.no_grad():
        want = ref(x, keep)
    ops.set_math_mode("bf16x3")
    try:
        with torch.no_grad():
            got = hip(x.to(DEV), keep.to(DEV)).cpu()
    finally:
        ops.set_math_mode("fp32")
    dropped = keep < 0.5
    assert torch.equal(got[dropped], x[dropped])                     # skipped tokens pass through bit-exactly
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


def test_token_skip_trunk_deit_s_shape():
    from laudnet_amd import ops
    from laudnet_amd.adavit import TokenSkipViT
    B, L, dim, heads, depth = 8, 197, 384, 6, 4
    ref = AR.TokenSkipViTRef(depth, dim, heads).eval()
    torch.manual_seed(11)
    for p_ in ref.parameters():
        if p_.dim() > 1:
            torch.nn.init.normal_(p_, std=0.03)
    hip = TokenSkipViT(depth, dim, heads).eval()
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV)
    x = seeded_randn((B, L, dim), 31)
    keeps = [_keep(B, L, 0.5, 40 + i) for i in range(depth)]
    with torch.no_grad():
        want = ref(x, keeps)
    ops.set_math_mode("bf16x3")
    try:
        with torch.no_grad():
            got = hip(x.to(DEV), [k.to(DEV) for k in keeps]).cpu()
    finally:
        ops.set_math_mode("fp32")
    assert (got - want).abs().max().item() < 1e-3 * max(1.0, want.abs().max().item())
