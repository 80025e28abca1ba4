"""Integer / index oracle (numpy).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, with plain integer arithmetic, the index semantics the reference gets
implicitly from torch ops on its masks:

* nearest up-sampling of the patch mask   -- laud_resnet.py:106 (F.interpolate nearest)
* adaptive average-pool bin edges         -- models/utils.py:48  (F.adaptive_avg_pool2d)
* ExpandMask dilation                     -- models/utils.py:74-89
* group -> channel ownership              -- models/utils.py:18-25

and defines the packed-index products the HIP path must reproduce bit-exactly
(row-major nonzero lists, 3x3 neighbour tables, per-image channel lists).
"""
from __future__ import annotations

import numpy as np


def nearest_src_index(out_size: int, in_size: int) -> np.ndarray:
    """Source index used by torch 'nearest' interpolation: min(floor(i*scale), in-1)
    with scale = float32(in)/float32(out) evaluated in float32 (ATen UpSample.h,
    nearest_neighbor_compute_source_index)."""
    scale = np.float32(in_size) / np.float32(out_size)
    i = np.arange(out_size, dtype=np.float32)
    src = np.floor(i * scale).astype(np.int64)
    return np.minimum(src, in_size - 1).astype(np.int32)


def adaptive_pool_bins(in_size: int, out_size: int):
    """[start,end) of each adaptive-avg-pool bin: floor(i*H/S), ceil((i+1)*H/S)."""
    i = np.arange(out_size, dtype=np.int64)
    start = (i * in_size) // out_size
    end = -((-(i + 1) * in_size) // out_size)
    return start.astype(np.int32), end.astype(np.int32)


def upsample_patch_mask(patch_mask: np.ndarray, out_h: int, out_w: int | None = None) -> np.ndarray:
    """[B,S,S] {0,1} -> [B,H,W] via nearest."""
    out_w = out_h if out_w is None else out_w
    ri = nearest_src_index(out_h, patch_mask.shape[1])
    ci = nearest_src_index(out_w, patch_mask.shape[2])
    return patch_mask[:, ri][:, :, ci]


def dilate_mask(mask: np.ndarray, stride: int, pad: int = 1) -> np.ndarray:
    """ExpandMask(stride, padding=pad) for one mask group: [B,H,W] bool -> [B,H*s,W*s] bool.
    Output pixel (y,x) is set iff some set input (i,j) has |y - i*s| <= pad and |x - j*s| <= pad."""
    b, h, w = mask.shape
    up = np.zeros((b, h * stride, w * stride), dtype=bool)
    up[:, ::stride, ::stride] = mask.astype(bool)
    if pad == 0:
        return up
    padded = np.pad(up, ((0, 0), (pad, pad), (pad, pad)))
    out = np.zeros_like(up)
    for dy in range(2 * pad + 1):
        for dx in range(2 * pad + 1):
            out |= padded[:, dy:dy + up.shape[1], dx:dx + up.shape[2]]
    return out


def nonzero_rows(mask: np.ndarray):
    """Row-major flat indices of set pixels over [B,H,W] plus the per-image prefix
    (length B+1).  Equivalent to torch.nonzero(mask.flatten())."""
    flat = np.flatnonzero(mask.reshape(-1)).astype(np.int32)
    per_img = mask.reshape(mask.shape[0], -1).sum(axis=1).astype(np.int64)
    prefix = np.concatenate([[0], np.cumsum(per_img)]).astype(np.int32)
    return flat, prefix


def position_map(mask: np.ndarray) -> np.ndarray:
    """Dense inverse of nonzero_rows: pos[b,y,x] = rank of the pixel in the packed
    list, -1 where the mask is 0."""
    flat_mask = mask.reshape(-1).astype(bool)
    pos = np.full(flat_mask.shape, -1, dtype=np.int32)
    pos[flat_mask] = np.arange(int(flat_mask.sum()), dtype=np.int32)
    return pos.reshape(mask.shape)


def neighbour_table(mask_out: np.ndarray, mask_in: np.ndarray, stride: int) -> np.ndarray:
    """For every set pixel of mask_out [B,Ho,Wo] (row-major), the packed row (w.r.t.
    mask_in [B,Hi,Wi]) of each of its 3x3/pad-1/stride-s input taps, tap = ky*3+kx;
    -1 for out-of-bounds or unset taps."""
    pos = position_map(mask_in)
    b_idx, oy, ox = np.nonzero(mask_out)
    hi, wi = mask_in.shape[1:]
    table = np.full((len(b_idx), 9), -1, dtype=np.int32)
    for ky in range(3):
        for kx in range(3):
            iy = oy * stride - 1 + ky
            ix = ox * stride - 1 + kx
            ok = (iy >= 0) & (iy < hi) & (ix >= 0) & (ix < wi)
            vals = np.full(len(b_idx), -1, dtype=np.int32)
            vals[ok] = pos[b_idx[ok], iy[ok], ix[ok]]
            table[:, ky * 3 + kx] = vals
    return table


def channel_lists(group_mask: np.ndarray, channels: int):
    """[B,G] {0,1} -> (idx [B,channels] int32 left-packed ascending, padded with -1;
    count [B] int32).  Group g owns channels [g*C/G, (g+1)*C/G)."""
    b, g = group_mask.shape
    per = channels // g
    full = np.repeat(group_mask.astype(bool), per, axis=1)
    idx = np.full((b, channels), -1, dtype=np.int32)
    cnt = full.sum(axis=1).astype(np.int32)
    for i in range(b):
        nz = np.flatnonzero(full[i])
        idx[i, :len(nz)] = nz
    return idx, cnt
