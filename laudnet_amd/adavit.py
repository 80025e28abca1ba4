"""Token skipping on the HIP path (BASELINE config 5: AdaViT / DeiT-S shaped blocks, "dynamic-token packed MHA").

The reference ships no model for this configuration, only the latency model of its operators
(DyNetSimulator/adavit/simulate_adavit.py:77-182): q / k / v, attention among the SELECTED tokens, projection /
MLP / residual updates on the selected tokens only (the latency model prices q / k / v on every token, :90-93; a dropped token is neither
query nor key, so they are computed for the attending tokens only -- TokenSkipBlock.qkv_kept_only).  This module executes that operator list on PACKED token lists with the
kernels of libldn_hip.so: the keep mask becomes a row list (ldn_mask_to_index on a [B, L, 1] mask), the linears are the packed-row
1x1 kernel (k_dense: gather rows in, scatter-add rows out, fused bias + residual), the attention is ldn_packed_mha (one workgroup
per image and head over the image's kept tokens).  HEAD skipping (a per-image head mask: q / k / v masked in the linear's epilogue, the
dropped heads' attention workgroups compute nothing) and LAYER skipping (per-image decisions for the attention and the MLP sub-block: the
image has no tokens in that sub-block's list) of simulate_adavit.py:81-88,140-182 ride on the same lists.  LayerNorm and GELU are epilogue terms of the linears (ldn_row_stats + the ln_* / relu-mode-3 arguments of ldn_conv_rows_split).  Parity is UNPINNED (there is
nothing in the reference to pin it to): tests compare against oracle/adavit_ref.py, a dense masked restatement of the same operator
list.  Inference only; no CPU fallback."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import LdnError


class TokenSkipBlock(nn.Module):
    """Pre-norm transformer block (DeiT layout: norm1, attn.qkv / proj, norm2, mlp.fc1 / fc2) that updates only the kept tokens."""

    def __init__(self, dim=384, heads=6, mlp_ratio=4.0):
        super().__init__()
        if dim % heads or dim // heads != 64:
            raise LdnError("TokenSkipBlock: head dimension must be 64 (DeiT / AdaViT: dim = 64 * heads)")
        hidden = int(dim * mlp_ratio)
        if dim % 32 or hidden % 32:
            raise LdnError("TokenSkipBlock: dim and the MLP width must be multiples of 32")
        self.dim, self.heads = dim, heads
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)
        self._w = None

    def _weights(self, dev):
        """Linear weights as k_dense wants them ([out, 1, in] fp32).  The two LayerNorms are folded into the linears that follow them
        (ldn_conv_rows_split: LN(x) . w + b = rstd (x . (gamma * w) - mean c1) + (w . beta + b)): qkv and fc1 carry gamma-scaled
        weights, c1 = their row sums and the constant as bias; only the rows' {mean, rstd} are computed per call (ldn_row_stats)."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._w is None or self._w[0] != key:
            with torch.no_grad():
                def plain(lin):
                    return (lin.weight.detach().float().reshape(lin.out_features, 1, lin.in_features).contiguous().to(dev),
                            lin.bias.detach().float().contiguous().to(dev))

                def after_ln(lin, ln):
                    w = lin.weight.detach().double()
                    g, be = ln.weight.detach().double(), ln.bias.detach().double()
                    wg = (w * g.view(1, -1)).float()
                    # c1 multiplies the row MEAN and cancels against the GEMM's large term when |mean| >> std: it must be the row sum
                    # of the weights the kernel actually multiplies with (the bf16 hi + lo split of wg, ~16 mantissa bits), not of
                    # the exact products -- the mismatch would be amplified by rstd
                    hi, lo = ops._hi_lo(wg)
                    c1 = (hi.double() + lo.double()).sum(dim=1)
                    return (wg.reshape(lin.out_features, 1, lin.in_features).contiguous().to(dev),
                            (w @ be + lin.bias.detach().double()).float().contiguous().to(dev), c1.float().contiguous().to(dev))
                self._w = (key, after_ln(self.qkv, self.norm1), plain(self.proj), after_ln(self.fc1, self.norm2), plain(self.fc2))
        return self._w[1:]

    qkv_kept_only = True      # class-level switch (A/B, tests): False = q / k / v for the rows of `qkv_rows` (every token of the running images)

    def run_packed(self, x2d, tok_rows, prefix, count, B, max_tokens, head_keep=None, mlp_lists=None, qkv_rows=None):
        """x2d [B*L, dim] fp32, updated IN PLACE on the kept tokens; tok_rows / prefix / count from ops.token_lists (the attention
        sub-block's token list).  Head and layer skipping (simulate_adavit.py:81-88,140-182), all optional:
          head_keep [B, heads] {0,1}  heads each image attends with: q / k / v of a dropped head are masked in the linear's epilogue,
                                      its attention workgroup computes nothing and contributes zeros to the projection;
          mlp_lists (rows, prefix, count)  the MLP sub-block's own token list (layer skipping decides the two sub-blocks separately:
                                      an image whose attention / MLP is skipped simply has no tokens in that list);
          qkv_rows (rows, count)      (only with qkv_kept_only = False) token rows q / k / v are computed for; by default they are
                                      computed for the attention list itself."""
        if self.training:
            raise LdnError("laudnet_amd implements the eval-mode (inference) hot path only")
        if ops.get_math_mode() != "bf16x3":
            raise LdnError("TokenSkipBlock runs in the bf16x3 arithmetic mode (ops.set_math_mode('bf16x3'))")
        (wq, bq, cq), (wp, bp), (w1, b1, c1), (w2, b2) = self._weights(x2d.device)
        rows = x2d.shape[0]
        Lt = rows // B
        q_rows, q_count = qkv_rows if qkv_rows is not None and not self.qkv_kept_only else (tok_rows, count)
        st = ops.row_stats(x2d, self.norm1.eps, rows=q_rows, count=q_count)                     # {mean, rstd} of the attending tokens (norm1)
        qkv = torch.empty(rows, 3 * self.dim, device=x2d.device, dtype=torch.float32)
        hk3 = None
        if head_keep is not None:     # [B, 3 * dim]: the head's decision over its 64 channels of q, k and v
            hk = head_keep.float().reshape(B, self.heads)
            hk3 = hk.repeat_interleave(self.dim // self.heads, dim=1).repeat(1, 3).contiguous()
        # norm1 + q / k / v of the tokens that attend (the attention list: a dropped token is neither query nor key,
        # simulate_adavit.py:96-109) -- not of every token: at keep 0.5 that is half of the widest linear of the block
        m_rows, _, m_count = mlp_lists if mlp_lists is not None else (tok_rows, prefix, count)
        # list lengths of this block's previous forward (pinned memory, no synchronisation): the tile-width hint of the row kernels
        hint = getattr(self, "_rows_hint", None)
        if hint is None:
            hint = self._rows_hint = ops.RowsHint(3)
        nq, na, nm = hint.get(0), hint.get(1), hint.get(2)
        hint.update(q_count, count, m_count)
        ops.conv_rows(x2d, wq, None, bq, qkv, a_rows=q_rows, out_rows=q_rows, taps=1, m_count=q_count, m_cap=rows, relu=0,
                      ln_stats=st, ln_c1=cq, chan_mask=hk3, rows_per_image=Lt if hk3 is not None else 0, rows_hint=nq)
        att = ops.packed_mha(qkv, tok_rows, prefix, B, self.heads, max_tokens,
                             head_keep=None if head_keep is None else head_keep.float().reshape(B, self.heads).contiguous())   # [capacity, dim], packed
        ops.conv_rows(att, wp, None, bp, x2d, taps=1, m_count=count, m_cap=rows, relu=0, out_rows=tok_rows, residual2d=x2d, rows_hint=na)
        st = ops.row_stats(x2d, self.norm2.eps, rows=m_rows, count=m_count)                     # norm2 (after the attention update)
        hid = torch.empty(rows, w1.shape[0], device=x2d.device, dtype=torch.float32)
        ops.conv_rows(x2d, w1, None, b1, hid, a_rows=m_rows, taps=1, m_count=m_count, m_cap=rows, relu=3, ln_stats=st, ln_c1=c1, rows_hint=nm)   # norm2 + fc1 + GELU
        ops.conv_rows(hid, w2, None, b2, x2d, taps=1, m_count=m_count, m_cap=rows, relu=0, out_rows=m_rows, residual2d=x2d, rows_hint=nm)
        return x2d

    @staticmethod
    def skip_lists(keep, attn_keep=None, mlp_keep=None):
        """Token lists of a block under token + layer skipping: keep [B, L] {0,1}; attn_keep / mlp_keep [B] {0,1} (None = run).
        -> (attention list (rows, prefix, count), MLP list or None, q/k/v rows (rows, count) or None)."""
        B, Lt = keep.shape
        ka = keep if attn_keep is None else keep * attn_keep.view(B, 1).to(keep.dtype)
        a_list = ops.token_lists(ka.contiguous())
        # the MLP sub-block has its OWN list whenever either decision is given: with attn_keep alone the attention list above has lost the
        # images whose attention is skipped, but their MLP still runs (mlp_keep None = run: the MLP list is then the plain token list)
        if mlp_keep is not None:
            m_list = ops.token_lists((keep * mlp_keep.view(B, 1).to(keep.dtype)).contiguous())
        else:
            m_list = None if attn_keep is None else ops.token_lists(keep.contiguous())
        q_rows = None
        if attn_keep is not None:
            r, _, c = ops.token_lists(attn_keep.view(B, 1).to(keep.dtype).expand(B, Lt).contiguous())
            q_rows = (r, c)
        return a_list, m_list, q_rows

    def forward(self, x, keep, head_keep=None, attn_keep=None, mlp_keep=None):
        """x [B, L, dim], keep [B, L] {0,1} -> new [B, L, dim] (kept tokens updated, the others passed through).  head_keep [B, heads],
        attn_keep / mlp_keep [B]: head and layer skipping (see run_packed)."""
        B, Lt, D = x.shape
        if Lt > 256:    # ldn_packed_mha holds at most 256 kept tokens of an image in LDS: more would be dropped silently
            raise LdnError("TokenSkipBlock: at most 256 tokens per image (ldn_packed_mha)")
        x2d = x.reshape(B * Lt, D).clone()
        (tok_rows, prefix, count), m_list, q_rows = self.skip_lists(keep, attn_keep, mlp_keep)
        self.run_packed(x2d, tok_rows, prefix, count, B, Lt, head_keep=head_keep, mlp_lists=m_list, qkv_rows=q_rows)
        return x2d.view(B, Lt, D)


class TokenSkipViT(nn.Module):
    """A trunk of token-skipping blocks (DeiT-S: depth 12, dim 384, 6 heads, MLP x4; 197 tokens).  forward(x, keeps): keeps[i] is the
    [B, L] keep mask of block i (the CLS token must be kept).  The residual stream is one [B*L, dim] buffer updated in place."""

    def __init__(self, depth=12, dim=384, heads=6, mlp_ratio=4.0):
        super().__init__()
        self.blocks = nn.ModuleList(TokenSkipBlock(dim, heads, mlp_ratio) for _ in range(depth))

    @staticmethod
    def blocks_qkv_kept_only():
        """Do the blocks compute q / k / v for the attending tokens only (the class-level switch of TokenSkipBlock)?"""
        return bool(TokenSkipBlock.qkv_kept_only)

    def forward(self, x, keeps, head_keeps=None, attn_keeps=None, mlp_keeps=None):
        """keeps[i] [B, L]; optional per-block head_keeps[i] [B, heads], attn_keeps[i] / mlp_keeps[i] [B] (head / layer skipping)."""
        B, Lt, D = x.shape
        if Lt > 256:
            raise LdnError("TokenSkipViT: at most 256 tokens per image (ldn_packed_mha)")
        x2d = x.reshape(B * Lt, D).clone()
        pick = lambda seq, i: None if seq is None else seq[i]
        for i, (blk, keep) in enumerate(zip(self.blocks, keeps)):
            (tok_rows, prefix, count), m_list, q_rows = blk.skip_lists(keep, pick(attn_keeps, i), pick(mlp_keeps, i))
            blk.run_packed(x2d, tok_rows, prefix, count, B, Lt, head_keep=pick(head_keeps, i), mlp_lists=m_list, qkv_rows=q_rows)
        return x2d.view(B, Lt, D)
