"""Token skipping on the HIP path (BASELINE config 5: AdaViT / DeiT-S shaped blocks, "dynamic-token packed MHA").

The reference ships no model for this configuration, only the latency model of its operators
(DyNetSimulator/adavit/simulate_adavit.py:77-182): q / k / v for every token, attention among the SELECTED tokens, projection /
MLP / residual updates on the selected tokens only.  This module executes that operator list on PACKED token lists with the
kernels of libldn_hip.so: the keep mask becomes a row list (ldn_mask_to_index on a [B, L, 1] mask), the linears are the packed-row
1x1 kernel (k_dense: gather rows in, scatter-add rows out, fused bias + residual), the attention is ldn_packed_mha (one workgroup
per image and head over the image's kept tokens).  The GELU is an epilogue of fc1 (relu mode 3 of ldn_conv_rows_split); LayerNorm is a library op.  Parity is UNPINNED (there is
nothing in the reference to pin it to): tests compare against oracle/adavit_ref.py, a dense masked restatement of the same operator
list.  Inference only; no CPU fallback."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

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
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._w is None or self._w[0] != key:
            f = lambda lin: (lin.weight.detach().float().reshape(lin.out_features, 1, lin.in_features).contiguous().to(dev),
                             lin.bias.detach().float().contiguous().to(dev))
            self._w = (key, f(self.qkv), f(self.proj), f(self.fc1), f(self.fc2))
        return self._w[1:]

    def run_packed(self, x2d, tok_rows, prefix, count, B, max_tokens):
        """x2d [B*L, dim] fp32, updated IN PLACE on the kept tokens; tok_rows / prefix / count from ops.token_lists."""
        if self.training:
            raise LdnError("laudnet_amd implements the eval-mode (inference) hot path only")
        if ops.get_math_mode() != "bf16x3":
            raise LdnError("TokenSkipBlock runs in the bf16x3 arithmetic mode (ops.set_math_mode('bf16x3'))")
        (wq, bq), (wp, bp), (w1, b1), (w2, b2) = self._weights(x2d.device)
        rows = x2d.shape[0]
        xn = F.layer_norm(x2d, (self.dim,), self.norm1.weight, self.norm1.bias, self.norm1.eps)
        qkv = torch.empty(rows, 3 * self.dim, device=x2d.device, dtype=torch.float32)
        ops.conv_rows(xn, wq, None, bq, qkv, taps=1, m_cap=rows, relu=0)                       # q / k / v for every token
        att = ops.packed_mha(qkv, tok_rows, prefix, B, self.heads, max_tokens)                  # [capacity, dim], packed
        ops.conv_rows(att, wp, None, bp, x2d, taps=1, m_count=count, m_cap=rows, relu=0, out_rows=tok_rows, residual2d=x2d)
        xn = F.layer_norm(x2d, (self.dim,), self.norm2.weight, self.norm2.bias, self.norm2.eps)
        hid = torch.empty(rows, w1.shape[0], device=x2d.device, dtype=torch.float32)
        ops.conv_rows(xn, w1, None, b1, hid, a_rows=tok_rows, taps=1, m_count=count, m_cap=rows, relu=3)   # fc1 + bias + GELU in the epilogue
        ops.conv_rows(hid, w2, None, b2, x2d, taps=1, m_count=count, m_cap=rows, relu=0, out_rows=tok_rows, residual2d=x2d)
        return x2d

    def forward(self, x, keep):
        """x [B, L, dim], keep [B, L] {0,1} -> new [B, L, dim] (kept tokens updated, the others passed through)."""
        B, Lt, D = x.shape
        if Lt > 256:    # ldn_packed_mha holds at most 256 kept tokens of an image in LDS: more would be dropped silently
            raise LdnError("TokenSkipBlock: at most 256 tokens per image (ldn_packed_mha)")
        x2d = x.reshape(B * Lt, D).clone()
        tok_rows, prefix, count = ops.token_lists(keep)
        self.run_packed(x2d, tok_rows, prefix, count, B, Lt)
        return x2d.view(B, Lt, D)


class TokenSkipViT(nn.Module):
    """A trunk of token-skipping blocks (DeiT-S: depth 12, dim 384, 6 heads, MLP x4; 197 tokens).  forward(x, keeps): keeps[i] is the
    [B, L] keep mask of block i (the CLS token must be kept).  The residual stream is one [B*L, dim] buffer updated in place."""

    def __init__(self, depth=12, dim=384, heads=6, mlp_ratio=4.0):
        super().__init__()
        self.blocks = nn.ModuleList(TokenSkipBlock(dim, heads, mlp_ratio) for _ in range(depth))

    def forward(self, x, keeps):
        B, Lt, D = x.shape
        if Lt > 256:
            raise LdnError("TokenSkipViT: at most 256 tokens per image (ldn_packed_mha)")
        x2d = x.reshape(B * Lt, D).clone()
        for blk, keep in zip(self.blocks, keeps):
            tok_rows, prefix, count = ops.token_lists(keep)
            blk.run_packed(x2d, tok_rows, prefix, count, B, Lt)
        return x2d.view(B, Lt, D)
