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


def test_token_skip_block_large_token_mean_and_wide_dim():
    """the LayerNorm fold rstd (x . w' - mean c1) cancels two large terms when |mean| >> std -- c1 is the row sum of
    the SPLIT weights the kernel multiplies with, so tokens with mean / std = 30 stay within 1e-3; dim 1280 (ViT-H: ldn_row_stats
    beyond 1024 columns) runs instead of raising."""
    from laudnet_amd import ops
    from laudnet_amd.adavit import TokenSkipBlock
    for dim, heads, off in ((384, 6, 30.0), (1280, 20, 3.0)):
        B, L = 2, 48
        ref = AR.TokenSkipBlockRef(dim, heads).eval()
        torch.manual_seed(9)
        for p_ in ref.parameters():
            if p_.dim() > 1:
                torch.nn.init.normal_(p_, std=0.05)
        hip = TokenSkipBlock(dim, heads).eval()
        hip.load_state_dict(ref.state_dict())
        hip = hip.to(DEV)
        x = seeded_randn((B, L, dim), 23)
        x[:, ::2] += off
        keep = _keep(B, L, 0.6, 24)
        with torch.no_grad():
            want = ref.double()(x.double(), keep.double()).float()
        ops.set_math_mode("bf16x3")
        try:
            with torch.no_grad():
                got = hip(x.to(DEV), keep.to(DEV)).cpu()
        finally:
            ops.set_math_mode("fp32")
        err = (got - want).abs().max().item()
        assert err < 1e-3 * max(1.0, want.abs().max().item()), (dim, err)


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


@pytest.mark.parametrize("rows,cin,cout,gather", [(2000, 384, 1152, False), (1500, 384, 1536, True), (300, 64, 128, True)])
def test_layernorm_and_gelu_as_epilogue_terms(rows, cin, cout, gather):
    """LN(x) . w + b with the LayerNorm applied AFTER the GEMM (ldn_row_stats + ln_stats / ln_c1 of ldn_conv_rows_split), and the exact
    GELU of relu mode 3, against torch's layer_norm -> linear -> gelu in fp64; rows with a large mean (cancellation in the folded
    form) included."""
    import torch.nn.functional as F
    from laudnet_amd import ops, load_library
    load_library()
    ops.set_math_mode("bf16x3")
    try:
        x = seeded_randn((rows, cin), 11)
        x[: rows // 4] += 3.0                                     # mean / std = 3: the folded form subtracts mean * c1 after the GEMM
        g, be = seeded_randn((cin,), 12) * 0.2 + 1.0, seeded_randn((cin,), 13) * 0.1
        w = seeded_randn((cout, cin), 14) * (1.0 / cin) ** 0.5
        b = seeded_randn((cout,), 15) * 0.1
        st = ops.row_stats(x.to(DEV), 1e-5)
        xd = x.double()
        mean, var = xd.mean(1), xd.var(1, unbiased=False)
        assert torch.allclose(st[:, 0].cpu().double(), mean, atol=1e-5) and torch.allclose(st[:, 1].cpu().double(), (var + 1e-5).rsqrt(), rtol=1e-5)
        wg = (w.double() * g.double().view(1, -1))
        src = torch.randperm(rows, generator=torch.Generator().manual_seed(3))[: rows // 2].to(torch.int32) if gather else None
        if gather:     # ldn_row_stats_list: the listed rows only (the first `count` entries), same floats; the others untouched
            lst = torch.cat((src, torch.zeros(7, dtype=torch.int32))).to(DEV)
            st2 = ops.row_stats(x.to(DEV), 1e-5, rows=lst, count=torch.tensor([rows // 2], dtype=torch.int32, device=DEV))
            assert torch.equal(st2[src.long().to(DEV)], st[src.long().to(DEV)])
            st = st2
        n = rows // 2 if gather else rows
        out = torch.zeros(n, cout, device=DEV)
        ops.conv_rows(x.to(DEV), wg.float().reshape(cout, 1, cin).contiguous().to(DEV), None, (w.double() @ be.double() + b.double()).float().to(DEV),
                      out, a_rows=None if src is None else src.to(DEV), taps=1, m_cap=n, relu=3, ln_stats=st, ln_c1=wg.sum(1).float().to(DEV))
        xs = xd if src is None else xd[src.long()]
        want = F.gelu(F.linear(F.layer_norm(xs, (cin,), g.double(), be.double(), 1e-5), w.double(), b.double()))
        err = (out.cpu().double() - want).abs().max().item()
        assert err < 2e-4, err
    finally:
        ops.set_math_mode(None)


@pytest.mark.parametrize("B,L,dim,heads", [(5, 40, 128, 2), (4, 197, 384, 6)])
def test_head_and_layer_skipping_vs_oracle(B, L, dim, heads):
    """Head skipping (per-image head masks) and layer skipping (per-image decisions for the attention and the MLP sub-block) of
    simulate_adavit.py:81-88,140-182 on top of token skipping, against the dense masked restatement (self-consistency: parity unpinned);
    images with no head / all heads, with one or both sub-blocks skipped (both: the image passes through bit-exactly)."""
    from laudnet_amd import ops
    from laudnet_amd.adavit import TokenSkipBlock, TokenSkipViT
    depth = 3
    ref = AR.TokenSkipViTRef(depth, dim, heads).eval()
    torch.manual_seed(13)
    for p_ in ref.parameters():
        if p_.dim() > 1:
            torch.nn.init.normal_(p_, std=0.05)
    hip = TokenSkipViT(depth, dim, heads).eval()
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV)
    x = seeded_randn((B, L, dim), 51)
    keeps = [_keep(B, L, 0.5, 60 + i) for i in range(depth)]
    hks, aks, mks = [], [], []
    for i in range(depth):
        hk = seeded_bernoulli((B, heads), 0.6, 70 + i)
        hk[0] = 0.0                                  # an image that drops every head
        hk[1] = 1.0                                  # ... keeps every head
        ak, mk = seeded_bernoulli((B,), 0.7, 80 + i), seeded_bernoulli((B,), 0.7, 90 + i)
        ak[2], mk[2] = 0.0, 0.0                      # image 2 skips both sub-blocks of every block
        ak[3], mk[3] = 1.0, 0.0
        hks.append(hk); aks.append(ak); mks.append(mk)
    with torch.no_grad():
        want = ref(x, keeps, hks, aks, mks)
    ops.set_math_mode("bf16x3")
    try:
        with torch.no_grad():
            got = hip(x.to(DEV), [k.to(DEV) for k in keeps], [h.to(DEV) for h in hks], [a.to(DEV) for a in aks], [m.to(DEV) for m in mks]).cpu()
            # one block, module-level call
            one = hip.blocks[0](x.to(DEV), keeps[0].to(DEV), head_keep=hks[0].to(DEV), attn_keep=aks[0].to(DEV), mlp_keep=mks[0].to(DEV)).cpu()
            want1 = ref.blocks[0](x, keeps[0], hks[0], aks[0], mks[0])
    finally:
        ops.set_math_mode("fp32")
    assert torch.equal(got[2], x[2])                                  # both sub-blocks skipped in every block: untouched
    scale = max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() < 2e-4 * scale, (got - want).abs().max().item()
    assert (one - want1).abs().max().item() < 1e-4 * max(1.0, want1.abs().max().item())


@pytest.mark.parametrize("which", ["attn_only", "mlp_only"])
def test_layer_skipping_with_one_decision_only(which):
    """attn_keep WITHOUT mlp_keep (and the reverse).  None means "the sub-block runs": an image whose attention is
    skipped still gets its MLP update (oracle/adavit_ref.py: km = keep when mlp_keep is None), and vice versa."""
    from laudnet_amd import ops
    from laudnet_amd.adavit import TokenSkipBlock
    B, L, dim, heads = 6, 40, 128, 2
    ref = AR.TokenSkipBlockRef(dim, heads).eval()
    torch.manual_seed(17)
    for p_ in ref.parameters():
        if p_.dim() > 1:
            torch.nn.init.normal_(p_, std=0.05)
    hip = TokenSkipBlock(dim, heads).eval()
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV)
    x = seeded_randn((B, L, dim), 52)
    keep = _keep(B, L, 0.5, 61)
    dec = seeded_bernoulli((B,), 0.5, 62)
    dec[0], dec[1] = 0.0, 1.0
    kw = dict(attn_keep=dec) if which == "attn_only" else dict(mlp_keep=dec)
    with torch.no_grad():
        want = ref(x, keep, None, kw.get("attn_keep"), kw.get("mlp_keep"))
    ops.set_math_mode("bf16x3")
    try:
        with torch.no_grad():
            got = hip(x.to(DEV), keep.to(DEV), **{k: v.to(DEV) for k, v in kw.items()}).cpu()
    finally:
        ops.set_math_mode("fp32")
    # image 0 skips one sub-block and must still be updated by the other
    assert not torch.equal(got[0], x[0])
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), (got - want).abs().max().item()
