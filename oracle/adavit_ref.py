"""ORACLE (test infrastructure only; PARITY UNPINNED): dense restatement of a token-skipping transformer block.

BASELINE config 5 ("AdaViT / DeiT-S token skipping") has NO model code in the reference: only the latency formulas of
DyNetSimulator/adavit/simulate_adavit.py:77-182 (the arithmetic lives in the un-vendored external repo MengLcool/AdaViT, no pinned
version).  What those formulas fix is the operator list of a block with token skipping:
  layernorm -> q / k / v linears (priced on EVERY token by the latency model, :90-93; only the selected tokens' q / k / v are ever
  used: this dense restatement computes all of them, the packed execution computes the selected ones -- same results) -> attention [B, heads, L_select, d] among the SELECTED tokens (:113-121)
  -> output projection on the selected tokens (:123-133) -> residual add on them (:169-171) -> layernorm -> fc1 / GELU / fc2 on the
  selected tokens (:136-150) -> residual add (:173-177); tokens that are not selected keep their value.  Head skipping (:81-88: attention
  over the selected heads of an image) and layer skipping (:140-182: the attention / MLP sub-block of an image runs or not) are per-image
  decisions: a dropped head's output is zero in front of the projection, a skipped sub-block leaves the image's tokens unchanged.
This file states exactly that, densely (a masked softmax over all tokens, results written only to the kept tokens), as the checker
of the packed execution in laudnet_amd/adavit.py.  It is a self-consistency oracle: no claim about AdaViT's own numerics."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class TokenSkipBlockRef(nn.Module):
    def __init__(self, dim=384, heads=6, mlp_ratio=4.0):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)

    def forward(self, x, keep, head_keep=None, attn_keep=None, mlp_keep=None):
        """x [B, L, dim]; keep [B, L] {0,1} (token 0 = CLS is always kept by the caller).  Head skipping (simulate_adavit.py:81-88):
        head_keep [B, heads] {0,1}, a dropped head contributes nothing to the projection's input.  Layer skipping (:140-182):
        attn_keep / mlp_keep [B] {0,1}, the image's tokens are not updated by a skipped sub-block."""
        B, L, D = x.shape
        h = self.heads
        q, k, v = self.qkv(self.norm1(x)).reshape(B, L, 3, h, D // h).permute(2, 0, 3, 1, 4)        # [B, h, L, d]
        s = (q @ k.transpose(-1, -2)) * (D // h) ** -0.5
        s = s.masked_fill(keep[:, None, None, :] < 0.5, float("-inf"))                              # only kept tokens are keys
        a = s.softmax(dim=-1) @ v                                                                   # [B, h, L, d]
        if head_keep is not None:
            a = a * head_keep.to(a.dtype)[:, :, None, None]
        a = a.transpose(1, 2).reshape(B, L, D)
        ka = keep if attn_keep is None else keep * attn_keep.to(keep.dtype)[:, None]
        km = keep if mlp_keep is None else keep * mlp_keep.to(keep.dtype)[:, None]
        x = x + ka[:, :, None] * self.proj(a)                                                       # only kept tokens are updated
        x = x + km[:, :, None] * self.fc2(F.gelu(self.fc1(self.norm2(x))))
        return x


class TokenSkipViTRef(nn.Module):
    """`depth` blocks; forward(x, keeps) with keeps [depth][B, L]."""

    def __init__(self, depth=12, dim=384, heads=6, mlp_ratio=4.0):
        super().__init__()
        self.blocks = nn.ModuleList(TokenSkipBlockRef(dim, heads, mlp_ratio) for _ in range(depth))

    def forward(self, x, keeps, head_keeps=None, attn_keeps=None, mlp_keeps=None):
        pick = lambda seq, i: None if seq is None else seq[i]
        for i, (blk, kp) in enumerate(zip(self.blocks, keeps)):
            x = blk(x, kp, pick(head_keeps, i), pick(attn_keeps, i), pick(mlp_keeps, i))
        return x
