"""Tensor-level wrappers over the C ABI (include/ldn_hip.h).  PyTorch is used only for device
memory and the current stream; every computation below runs in libldn_hip.so.  No fallbacks."""
from __future__ import annotations

import os
import threading
import weakref
from dataclasses import dataclass

import torch

from . import _lib as L


def _f32c(t, what):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise L.LdnError(f"{what}: expected a contiguous float32 tensor, got {t.dtype} strides {t.stride()}")
    return t


def _f32rows(t, what):
    """2-D fp32 tensor whose rows are contiguous (a column slice of a wider matrix is fine: the row stride is passed on)."""
    if t is None:
        return None
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.data_ptr() % 16 != 0:
        raise L.LdnError(f"{what}: expected a 16-byte aligned float32 matrix with contiguous rows, got {t.dtype} strides {t.stride()}")
    return t


def _i32c(t, what):
    if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
        raise L.LdnError(f"{what}: expected a contiguous int32 tensor")
    return t


MATH_MODES = {"fp32": 0, "bf16x3": 1}
_math = threading.local()   # the host-side default; the C library itself is stateless (math_mode is an argument of every conv call)


def set_math_mode(mode):
    """Default arithmetic of the MFMA convolutions issued from THIS thread: "fp32", "bf16x3" or None (= the library's
    read-only process default, env LDN_MATH_MODE).  Passed to libldn_hip.so as the math_mode argument of each call."""
    if mode is not None and mode not in MATH_MODES:
        raise L.LdnError(f"set_math_mode: unknown mode {mode!r} (expected one of {sorted(MATH_MODES)})")
    _math.mode = mode


def get_math_mode():
    mode = getattr(_math, "mode", None)
    if mode is None:
        m = L.load().ldn_default_math_mode()
        mode = next(k for k, v in MATH_MODES.items() if v == m)
    return mode


def _mm(math=None):
    mode = math if math is not None else getattr(_math, "mode", None)
    return -1 if mode is None else MATH_MODES[mode]


def _work(nbytes, dev):
    return torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32) if nbytes else None


def as_nhwc(x):
    """[B,C,H,W] logical tensor -> contiguous [B,H,W,C] view (copying into channels_last if needed)."""
    if x.dim() != 4:
        raise L.LdnError("expected a 4-D NCHW tensor")
    if x.dtype != torch.float32:
        raise L.LdnError("laudnet_amd computes in float32")
    x = x.contiguous(memory_format=torch.channels_last)
    v = x.permute(0, 2, 3, 1)
    if not v.is_contiguous():  # degenerate strides (C==1 or H*W==1): materialise
        v = v.contiguous()
    return v


def from_nhwc(y):
    """contiguous [B,H,W,C] -> [B,C,H,W] logical view (channels_last memory, no copy)."""
    return y.permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------- a1
def spatial_masker(x_nhwc, weight, bias, groups, mask_size, want_logits=False, carry=None, return_work=False):
    """Masker_spatial eval forward (models/utils.py:47-65).  x_nhwc [B,H,W,C]; weight [2g,C]; bias [2g].
    Returns (mask [B,g,Sy,Sx] float {0,1}, logits [B,2g,Sy,Sx] or None[, work]).  carry = (work, prefix, shape_key[, patch_mask]) of the
    previous block on the same residual stream (see ldn_spatial_masker): images (layer skip) / patches (patch masks) that block left
    untouched are not re-read."""
    L.require_device(x_nhwc, weight, bias)
    lib = L.load()
    B, H, W, C = x_nhwc.shape
    pooled = mask_size < H
    sy, sx = (mask_size, mask_size) if pooled else (H, W)
    mask = torch.empty(B, groups, sy, sx, device=x_nhwc.device, dtype=torch.float32)
    logits = torch.empty(B, 2 * groups, sy, sx, device=x_nhwc.device, dtype=torch.float32) if want_logits else None
    nbytes = lib.ldn_spatial_masker_workspace_bytes(B, H, W, C, mask_size)
    prefix = cmask = None
    # a carry is only taken from a masker call on a tensor of the SAME shape (the partial sums are [B][splits][C] of that shape: a
    # byte count alone can coincide between a stride-2 block's input and the next block's input)
    shape_key = (B, H, W, C, mask_size)
    if (carry is not None and nbytes and carry[0] is not None and len(carry) > 2 and carry[2] is not None
            and tuple(carry[2]) == shape_key and carry[0].numel() * 4 == nbytes and carry[0].device == x_nhwc.device):
        work = carry[0]
        if mask_size == 1:
            prefix = _i32c(carry[1], "carry_prefix")
        else:   # patch masks: carry[3] = the [B, S, S] patch mask the previous block executed
            cmask = _f32c(carry[3], "carry_mask") if len(carry) > 3 and carry[3] is not None else None
            # uneven grids (H % S != 0): adaptive-pool bins overlap the nearest-mapped regions of neighbouring patches, so an
            # untouched patch's stored means can be stale -- no carry there (the library refuses it too)
            if cmask is None or tuple(cmask.shape) != (B, sy, sx) or H % mask_size or W % mask_size:
                work, cmask = _work(nbytes, x_nhwc.device), None
    else:
        work = _work(nbytes, x_nhwc.device) if (mask_size == 1 or return_work) else None
    L.check(lib.ldn_spatial_masker(L.ptr(_f32c(x_nhwc, "x")), B, H, W, C, L.ptr(_f32c(weight, "w")),
                                   L.ptr(_f32c(bias, "bias")), groups, mask_size, L.ptr(mask), L.ptr(logits), L.ptr(work),
                                   L.ptr(prefix), L.ptr(cmask), L.stream_ptr()), "ldn_spatial_masker")
    if return_work and work is not None:
        work.ldn_shape_key = shape_key          # what a later call must match to reuse these sums (carry[2])
    return (mask, logits, work) if return_work else (mask, logits)


# ---------------------------------------------------------------------------------------- a4/a11
@dataclass
class IndexSet:
    idx3: torch.Tensor
    pos3: torch.Tensor
    idx1: torch.Tensor
    pos1: torch.Tensor
    nbr: torch.Tensor
    cnt: torch.Tensor      # [2] = {#rows of mask3, #rows of mask1}
    pre3: torch.Tensor     # [B+1]
    pre1: torch.Tensor     # [B+1]
    stats: torch.Tensor    # [3] = {mean patch mask, mean mask2, mean mask1}
    cap3: int
    cap1: int
    patch_major: bool = False     # idx3 lists the kept pixels patch by patch (ldn_mask_plan): whole patches are consecutive packed rows
    work_vouched: bool = False    # the list build's work buffer is the shared zeroed one (_vouched_call)


def _empty_index(B, out_h, out_w, stride, dev, plan_S=None):
    lib = L.load()
    cap3, cap1 = B * out_h * out_w, B * out_h * stride * out_w * stride
    i32 = dict(device=dev, dtype=torch.int32)
    ix = IndexSet(idx3=torch.empty(cap3, **i32), pos3=torch.empty(cap3, **i32), idx1=torch.empty(cap1, **i32),
                  pos1=torch.empty(cap1, **i32), nbr=torch.empty(cap3 * 9, **i32), cnt=torch.empty(2, **i32),
                  pre3=torch.empty(B + 1, **i32), pre1=torch.empty(B + 1, **i32),
                  stats=torch.empty(3, device=dev, dtype=torch.float32), cap3=cap3, cap1=cap1)
    nwork = max(lib.ldn_mask_to_index_workspace_bytes(B, out_h, out_w, stride) // 4, 1)
    if plan_S is not None and USE_CLEAN_PLAN_WORK and _plan_path(lib, plan_S[0], plan_S[1], out_h, out_w, stride):
        # the one-launch build leaves its flag words zero: a buffer zeroed ONCE per (device, stream) serves every build (ldn_plan_work_zeroed)
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        work = _PLAN_WORK.get(key)
        if work is None or work.numel() < nwork:
            work = _PLAN_WORK[key] = torch.zeros(max(nwork, 4096), **i32)
        ix.work_vouched = True      # the caller arms ldn_plan_work_zeroed right in front of its library call (_vouched_call)
        return ix, work
    return ix, torch.empty(nwork, **i32)


def _vouched_call(ix, fn):
    """Run fn() -- ONE list-build call of the library -- with `ldn_plan_work_zeroed(1)` armed when ix's work buffer is the shared zeroed one.  The
    flag is per-thread state that the call consumes; an exception raised before the call happens must not leave it armed for an unrelated later
    call, hence the finally."""
    vouched = getattr(ix, "work_vouched", False)
    lib = L.load()
    if vouched:
        lib.ldn_plan_work_zeroed(1)
    try:
        return fn()
    finally:
        if vouched:
            lib.ldn_plan_work_zeroed(0)


_PLAN_WORK = {}
USE_CLEAN_PLAN_WORK = os.environ.get("LDN_PLAN_CLEAN_WORK", "1") != "0"    # tuning switch (A/B): "0" = a fresh work buffer + a zeroing launch per list build


def _atoi(text):
    """C's atoi: leading blanks, an optional sign, digits; 0 when there are none."""
    import re
    m = re.match(r"\s*([+-]?\d+)", text)
    return int(m.group(1)) if m else 0


def _plan_path(lib, S, Sx, out_h, out_w, stride):
    """Will ldn_mask_plan / ldn_mask_to_index build these lists in ONE launch (k_plan)?  (mirrors the library's own choice)"""
    plan_env = os.environ.get("LDN_INDEX_PLAN")
    if "LDN_INDEX_BANDS" in os.environ or (plan_env is not None and _atoi(plan_env) == 0):      # (the library's own tests: getenv != NULL, atoi == 0)
        return False
    return bool(lib.ldn_mask_plan_fits(int(S), int(Sx), int(out_h), int(out_w), int(stride)))


def coarsen_cell_means(fine, S):
    """fine [B, 2S, 2S, C] cell means -> [B, S, S, C] means of the 2 x 2 groups of cells (ldn_coarsen_cell_means)."""
    L.require_device(fine)
    B, S2, S2x, C = fine.shape
    if S2 != 2 * S or S2x != 2 * S:
        raise L.LdnError("coarsen_cell_means: fine must be [B, 2S, 2S, C]")
    coarse = torch.empty(B, S, S, C, device=fine.device, dtype=torch.float32)
    L.check(L.load().ldn_coarsen_cell_means(L.ptr(_f32c(fine, "fine")), B, S, C, L.ptr(coarse), L.stream_ptr()), "ldn_coarsen_cell_means")
    return coarse


def plan_timeouts(reset=False, raise_on_error=False):
    """Prefix waits of the one-launch list build that ran into their time bound since the last reset (ldn_plan_timeouts; 0 on a
    healthy device; such a launch leaves EMPTY lists, never uninitialised ones).  Synchronises the device."""
    import ctypes
    n = ctypes.c_int(0)
    L.check(L.load().ldn_plan_timeouts(ctypes.byref(n), 1 if reset else 0), "ldn_plan_timeouts", fault_ok=True)
    if raise_on_error and n.value:
        raise L.LdnError(f"{n.value} list-build launches ran into their time bound: their blocks saw empty pixel lists (results invalid)")
    return n.value


def mask_plan_fits(S, Sx, out_h, out_w, stride=1):
    """Whether the one-launch list build (ldn_mask_plan) holds a map of this size (per-image tables in one workgroup's LDS)."""
    return bool(L.load().ldn_mask_plan_fits(S, Sx, out_h, out_w, stride))


def mask_to_index(patch_mask, out_h, out_w, stride, patch_major=False):
    """patch_mask [B,Sy,Sx] float {0,1} -> packed index lists (see ldn_mask_to_index).  patch_major: the kept pixels patch by
    patch instead of row-major over the image (ldn_mask_plan; even grids, maps that fit one workgroup's tables)."""
    L.require_device(patch_mask)
    lib = L.load()
    if patch_mask.dim() != 3:
        raise L.LdnError("mask_to_index: patch_mask must be [B,Sy,Sx]")
    B, S, Sx = patch_mask.shape
    # (S == Sx == 1 is the closed-form layer-skip build: it does not touch `work`)
    ix, work = _empty_index(B, out_h, out_w, stride, patch_mask.device, plan_S=(S, Sx) if (patch_major or S > 1 or Sx > 1 or os.environ.get("LDN_INDEX_GENERIC")) else None)
    if patch_major:
        ix.patch_major = True
        pm = _f32c(patch_mask, "patch_mask")
        L.check(_vouched_call(ix, lambda: lib.ldn_mask_plan(L.ptr(pm), None, 0, None, None, None, None, B, S, Sx, out_h, out_w,
                                  stride, 1, L.ptr(ix.idx3), L.ptr(ix.pos3), L.ptr(ix.idx1), L.ptr(ix.pos1), L.ptr(ix.nbr),
                                  L.ptr(ix.cnt), L.ptr(ix.pre3), L.ptr(ix.pre1), L.ptr(ix.stats), L.ptr(work), L.stream_ptr())),
                "ldn_mask_plan")
        return ix
    pm = _f32c(patch_mask, "patch_mask")
    L.check(_vouched_call(ix, lambda: lib.ldn_mask_to_index(L.ptr(pm), B, S, Sx, out_h, out_w, stride, L.ptr(ix.idx3),
                                  L.ptr(ix.pos3), L.ptr(ix.idx1), L.ptr(ix.pos1), L.ptr(ix.nbr), L.ptr(ix.cnt),
                                  L.ptr(ix.pre3), L.ptr(ix.pre1), L.ptr(ix.stats), L.ptr(work), L.stream_ptr())),
            "ldn_mask_to_index")
    return ix


def layer_index(image_mask, out_h, out_w, stride=1, tile=None):
    """Layer skip: image_mask [B] float {0,1} -> packed lists (ldn_layer_index); tile = (gy, gx): the kept images' pixels tile by tile
    (IndexSet.patch_major), None: row-major (= mask_to_index of a [B,1,1] mask)."""
    L.require_device(image_mask)
    lib = L.load()
    B = image_mask.numel()
    ix, _ = _empty_index(B, out_h, out_w, stride, image_mask.device)
    gy, gx = tile if tile is not None else (0, 0)
    ix.patch_major = tile is not None
    L.check(lib.ldn_layer_index(L.ptr(_f32c(image_mask.reshape(B), "image_mask")), B, out_h, out_w, stride, gy, gx, L.ptr(ix.idx3),
                                L.ptr(ix.pos3), L.ptr(ix.idx1), L.ptr(ix.pos1), L.ptr(ix.nbr), L.ptr(ix.cnt), L.ptr(ix.pre3),
                                L.ptr(ix.pre1), L.ptr(ix.stats), L.stream_ptr()), "ldn_layer_index")
    return ix


def layer_head(pool, weight, bias, groups, want_logits=False):
    """The layer-skip decision from pooled tile means (ldn_layer_head): pool [B,nparts,C] -> (mask [B,g,1,1], logits [B,2g,1,1] or None)."""
    L.require_device(pool, weight, bias)
    lib = L.load()
    B, nparts, C = pool.shape
    mask = torch.empty(B, groups, 1, 1, device=pool.device, dtype=torch.float32)
    logits = torch.empty(B, 2 * groups, 1, 1, device=pool.device, dtype=torch.float32) if want_logits else None
    L.check(lib.ldn_layer_head(L.ptr(_f32c(pool, "pool")), B, nparts, C, L.ptr(_f32c(weight, "w")), L.ptr(_f32c(bias, "bias")), groups,
                               L.ptr(mask), L.ptr(logits), L.stream_ptr()), "ldn_layer_head")
    return mask, logits


def mask_plan(pool, weight, bias, out_h, out_w, stride=1, patch_major=True, want_logits=False):
    """The fused spatial masker + list build of a block whose input's pooled patch means are already known (ldn_mask_plan, decide
    mode): pool [B,S,Sx,C] (the `work` of spatial_masker, refreshed by the previous block's conv_rows(..., pool=...)), weight [2,C],
    bias [2] (one mask group).  Returns (mask [B,1,S,Sx], logits [B,2,S,Sx] or None, IndexSet); x is not read."""
    L.require_device(pool, weight, bias)
    lib = L.load()
    if pool.dim() != 4 or tuple(weight.shape) != (2, pool.shape[3]) or bias.numel() != 2:
        raise L.LdnError("mask_plan: pool must be [B,S,Sx,C], weight [2,C], bias [2] (one mask group)")
    B, S, Sx, C = pool.shape
    dev = pool.device
    ix, work = _empty_index(B, out_h, out_w, stride, dev, plan_S=(S, Sx))
    ix.patch_major = bool(patch_major)
    mask = torch.empty(B, 1, S, Sx, device=dev, dtype=torch.float32)
    logits = torch.empty(B, 2, S, Sx, device=dev, dtype=torch.float32) if want_logits else None
    pl, wt, bs = _f32c(pool, "pool"), _f32c(weight, "w"), _f32c(bias, "bias")
    L.check(_vouched_call(ix, lambda: lib.ldn_mask_plan(None, L.ptr(pl), C, L.ptr(wt), L.ptr(bs), L.ptr(mask),
                              L.ptr(logits), B, S, Sx, out_h, out_w, stride, 1 if patch_major else 0, L.ptr(ix.idx3), L.ptr(ix.pos3),
                              L.ptr(ix.idx1), L.ptr(ix.pos1), L.ptr(ix.nbr), L.ptr(ix.cnt), L.ptr(ix.pre3), L.ptr(ix.pre1),
                              L.ptr(ix.stats), L.ptr(work), L.stream_ptr())), "ldn_mask_plan")
    return mask, logits, ix


# ---------------------------------------------------------------------------------------- K2/K5
def gather_rows(src2d, rows, count=None, cap=None):
    L.require_device(src2d, rows)
    lib = L.load()
    cap = rows.numel() if cap is None else cap
    C = src2d.shape[1]
    out = torch.empty(cap, C, device=src2d.device, dtype=torch.float32)
    L.check(lib.ldn_gather_rows(L.ptr(_f32c(src2d, "src")), src2d.stride(0), L.ptr(_i32c(rows, "rows")),
                                L.ptr(_i32c(count, "count")), cap, C, L.ptr(out), C, L.stream_ptr()), "ldn_gather_rows")
    return out


def scatter_add_relu(packed, rows, identity2d, out2d=None, count=None, cap=None):
    L.require_device(packed, rows, identity2d)
    lib = L.load()
    cap = rows.numel() if cap is None else cap
    out2d = identity2d if out2d is None else out2d
    C = packed.shape[1]
    L.check(lib.ldn_scatter_add_relu(L.ptr(_f32c(packed, "packed")), packed.stride(0), L.ptr(_i32c(rows, "rows")),
                                     L.ptr(_i32c(count, "count")), cap, C, L.ptr(_f32c(identity2d, "identity")),
                                     identity2d.stride(0), L.ptr(_f32c(out2d, "out")), out2d.stride(0),
                                     L.stream_ptr()), "ldn_scatter_add_relu")
    return out2d


def forward_stats(terms, static_flops, *, cnt=None, denom=None, st_in=None):
    """The FLOPs bookkeeping of a forward in one launch (see ldn_forward_stats).  terms [n,5] float64 (device); cnt [n,B] int32 +
    denom [n] float32 (channel-mode blocks: cs = cnt.sum() / denom where denom > 0) and / or st_in [n,3|4] float32.
    Returns (st [n,4] = s3, s2, s1, cs; perc [n]; flops 0-dim)."""
    L.require_device(terms, cnt, denom, st_in)
    lib = L.load()
    n = terms.shape[0]
    dev = terms.device
    if terms.dtype != torch.float64 or tuple(terms.shape) != (n, 5) or not terms.is_contiguous():
        raise L.LdnError("forward_stats: terms must be a contiguous float64 [n, 5] tensor")
    if cnt is not None and (tuple(cnt.shape[:1]) != (n,) or cnt.dim() != 2 or denom is None or tuple(denom.shape) != (n,)):
        raise L.LdnError("forward_stats: cnt must be [n, B] with denom [n]")
    if st_in is not None and (st_in.dim() != 2 or st_in.shape[0] != n or st_in.shape[1] not in (3, 4)):
        raise L.LdnError("forward_stats: st_in must be [n, 3] or [n, 4]")
    st = torch.empty(n, 4, device=dev, dtype=torch.float32)
    perc = torch.empty(n, device=dev, dtype=torch.float32)
    flops = torch.empty((), device=dev, dtype=torch.float32)
    L.check(lib.ldn_forward_stats(L.ptr(_i32c(cnt, "cnt")), cnt.shape[1] if cnt is not None else 0, L.ptr(_f32c(denom, "denom")),
                                  L.ptr(_f32c(st_in, "st_in")), st_in.shape[1] if st_in is not None else 0, L.ptr(terms),
                                  float(static_flops), n, L.ptr(st), L.ptr(perc), L.ptr(flops), L.stream_ptr(terms)), "ldn_forward_stats")
    return st, perc, flops


# ---------------------------------------------------------------------------------------- a7 rows
_SPLIT_CACHE = {}   # (data_ptr, _version, shape) of an fp32 weight -> its pre-split n-major copy (ldn_conv_rows_split)
DENSE_TAPS = tuple(int(t) for t in os.environ.get("LDN_DENSE_TAPS", "1,9").split(","))   # tuning: "1,9" sends the packed-row 3x3 to k_dense too
DENSE_CHANNEL_3X3 = os.environ.get("LDN_DENSE_CHANNEL_3X3", "1") != "0"   # the 3x3 of the dense channel execution (stage 4's first block) on k_dense's neighbour-table form, mask and post-ReLU constant in its epilogue (round 2: no gain; round 4: -0.05 ms per step, no separate mask pass)
DENSE_K_MULT = int(os.environ.get("LDN_DENSE_K_MULT", "8"))   # tuning: 32 keeps layers whose widths are not multiples of 32 (LAD-RegNet 144 / 784) on the round-1 kernels
DENSE_N_MULT = 4 if DENSE_K_MULT == 8 else 32
USE_DENSE_KERNEL = os.environ.get("LDN_DENSE_KERNEL", "1") != "0"   # tuning switch: off keeps every packed-row 1x1 on the round-1 kernels
# fp32 math mode on k_dense's true-fp32 form (ldn_conv_rows_f32)?  Measured (round 4, bs256, one box): RegNet 7.12 -> 6.88 ms, but spatial
# 27.5 -> 32.6 and layer 25.1 -> 29.0 ms: at fp32 MFMA rates every tile is matrix-bound, so the 256-row tiles' under-filled grids (98-196
# workgroups on 256 CUs at stage 3) cost their full share while round 1's smaller tiles fill the chip -- off by default, kept as the
# ABI's fp32 entry point with the epilogue features (channel mask, shift table, LayerNorm / GELU terms) round 1's kernel lacks
USE_DENSE_F32 = os.environ.get("LDN_DENSE_F32", "0") != "0"
# fp32 math mode on the fused channel-mode kernels (k_head / k_tail / k_chain in true-fp32 MFMA arithmetic: ldn_bottleneck_*_f32)
USE_FUSED_F32 = os.environ.get("LDN_FUSED_F32", "1") != "0"


def dense_kernel_ok():
    """Does the current arithmetic mode run the shared-weight row convolutions on k_dense (bf16x3: ldn_conv_rows_split; fp32:
    ldn_conv_rows_f32)?"""
    mode = get_math_mode()
    return USE_DENSE_KERNEL and (mode == "bf16x3" or (mode == "fp32" and USE_DENSE_F32))


def fused_kernel_ok():
    """Does the current arithmetic mode run channel-mode blocks on the fused kernels (k_head / k_tail / k_chain: bf16x3, or their
    true-fp32 forms ldn_bottleneck_*_f32)?"""
    mode = get_math_mode()
    return mode == "bf16x3" or (mode == "fp32" and USE_FUSED_F32)


def split_rows_weight(w):
    """[cout, 1, cin] (or [cout, cin]) fp32 -> [cout][cin/8][8 hi | 8 lo] bf16, cached until the tensor changes."""
    ident = (w.data_ptr(), tuple(w.shape), str(w.device))
    key = ident + (w._version,)
    hit = _SPLIT_CACHE.get(key)
    if hit is None:
        # a long-lived weight edited in place (BN re-estimation, fine-tuning loops that call the eval path) bumps _version at every
        # edit: drop the copies of its older versions, or the cache grows by one entry per edit until the tensor dies
        for stale in [k for k in _SPLIT_CACHE if k[:3] == ident]:
            _SPLIT_CACHE.pop(stale, None)
        with torch.no_grad():
            w2 = w.detach().float().reshape(w.shape[0], -1)
            hit = _SPLIT_CACHE[key] = (pack_w1_split(w2),)
        # the entry lives exactly as long as its source tensor (a module's folded weight dropped by invalidate() / a new checkpoint
        # takes its split copy with it; a later tensor that reuses the address starts from an empty slot)
        # (tied to the tensor that OWNS the storage: callers pass throw-away views such as w3[channel_slice])
        weakref.finalize(w._base if w._base is not None else w, _SPLIT_CACHE.pop, key, None)
    return hit[0]


def row_stats(x2d, eps=1e-5, rows=None, count=None):
    """LayerNorm statistics of the rows of x2d [rows, C]: [rows, 2] = {mean, 1 / sqrt(biased var + eps)} (see ldn_row_stats).
    rows / count: only the listed rows (ldn_row_stats_list; the other entries of the result are uninitialised)."""
    L.require_device(x2d)
    st = torch.empty(x2d.shape[0], 2, device=x2d.device, dtype=torch.float32)
    if rows is not None:
        L.check(L.load().ldn_row_stats_list(L.ptr(_f32rows(x2d, "x")), x2d.stride(0), x2d.shape[0], x2d.shape[1], float(eps),
                                            L.ptr(_i32c(rows, "rows")), L.ptr(_i32c(count, "count")), L.ptr(st), L.stream_ptr(x2d)),
                "ldn_row_stats_list")
        return st
    L.check(L.load().ldn_row_stats(L.ptr(_f32rows(x2d, "x")), x2d.stride(0), x2d.shape[0], x2d.shape[1], float(eps), L.ptr(st),
                                   L.stream_ptr(x2d)), "ldn_row_stats")
    return st


class RowsHint:
    """Row counts of the PREVIOUS forward for the tile-width choice of k_dense (ldn_hint_rows): the device-side counts of a block are
    copied to pinned memory without a synchronisation (update), the next forward reads what has arrived (get).  A stale or missing
    value only changes tile shapes, never results."""

    EVERY = 32      # after the first two forwards the counts are refreshed on every 32nd call only (a copy is a launch on the stream)

    def __init__(self, n):
        self.n, self.host, self.seen, self.calls = n, None, False, 0

    def get(self, i):
        if not self.seen or self.host is None:
            return None
        v = int(self.host[i])
        return v if v >= 0 else None

    def update(self, *dev_counts):
        """dev_counts: int32 device tensors whose elements fill the slots in order."""
        self.calls += 1
        if (self.calls > 2 and self.calls % self.EVERY) or torch.cuda.is_current_stream_capturing():
            return
        if self.host is None:
            self.host = torch.full((self.n,), -1, dtype=torch.int32).pin_memory()
        o = 0
        for t in dev_counts:
            k = t.numel()
            self.host[o:o + k].copy_(t.reshape(-1), non_blocking=True)
            o += k
        self.seen = True


USE_ROWS_HINT = os.environ.get("LDN_ROWS_HINT", "1") != "0"


def conv_rows(a2d, w, scale, shift, out2d, *, a_rows=None, taps=1, m_count=None, m_cap=None, relu=1, rows_hint=None,
              relu_if_neg=None, out_rows=None, residual2d=None, math=None, post_sub=None, chan_mask=None, rows_per_image=0,
              pix_map=None, geom=None, ln_stats=None, ln_c1=None, pool=None, pool_grid=None):
    """Packed-row convolution (see ldn_conv_rows).  a2d [rows,lda>=cin]; w [cout,taps,cin]; out2d [rows,ldo].
    In bf16x3 mode the 1x1 form runs on k_dense (ldn_conv_rows_split) with a cached pre-split copy of the weights;
    post_sub / chan_mask (dense execution of channel mode) exist on that kernel only."""
    L.require_device(a2d, w, out2d)
    lib = L.load()
    cout, t, cin = w.shape
    if t != taps:
        raise L.LdnError(f"conv_rows: weight has {t} taps, expected {taps}")
    if m_cap is None:
        m_cap = a2d.shape[0] if a_rows is None else a_rows.numel() // taps
    mode = math if math is not None else get_math_mode()
    # k_dense also implements the 3x3 over a neighbour table (taps == 9).  Before its weight fragments were double-buffered it was
    # SLOWER there than round 1's producer/consumer kernel (spatial 17.8 -> 19.1 ms); with the pinned schedule it is faster on the
    # packed-row paths (spatial 16.62 -> 16.31 ms, same box) and neutral on the dense channel execution of stage 4, which stays on
    # k_conv_bf3 (DENSE_CHANNEL_3X3).  LDN_DENSE_TAPS=1 restores the old dispatch.
    # round 4: the same kernel in true-fp32 MFMA arithmetic (ldn_conv_rows_f32: plain fp32 weights, no split copy) -- the fp32 math mode
    # no longer falls back to round 1's producer / consumer kernel on the shared-weight row paths
    dense_ok = (USE_DENSE_KERNEL and (mode == "bf16x3" or (mode == "fp32" and USE_DENSE_F32)) and taps in DENSE_TAPS
                and cin % DENSE_K_MULT == 0 and cout % DENSE_N_MULT == 0 and a2d.stride(0) >= cin)
    classes = 1 if shift.dim() == 1 else shift.shape[0]
    if (post_sub is not None or chan_mask is not None or classes != 1 or relu == 3 or ln_stats is not None) and not dense_ok:
        raise L.LdnError("conv_rows: post_sub / chan_mask / a shift table / the GELU and LayerNorm epilogues need the k_dense path (cin % 8 == 0, cout % 4 == 0)")
    if rows_hint is not None and m_count is not None and USE_ROWS_HINT and dense_ok:
        lib.ldn_hint_rows(int(rows_hint))
    if pool is not None:
        # the pooled patch means of the output as a by-product (ldn_conv_rows_pool): pool [B,S,Sx,cout], pool_grid = (S, Sx, Ho, Wo);
        # the packed rows list whole patches (IndexSet.patch_major)
        if not dense_ok or taps != 1 or post_sub is not None or chan_mask is not None or ln_stats is not None or classes != 1:
            raise L.LdnError("conv_rows: pool= needs the 1x1 k_dense form without post_sub / chan_mask / LayerNorm terms")
        S, Sx, ho, wo = pool_grid
        wptr = split_rows_weight(w) if mode == "bf16x3" else _f32c(w, "w")
        L.check(lib.ldn_conv_rows_pool(L.ptr(_f32rows(a2d, "a")), a2d.stride(0), L.ptr(_i32c(a_rows, "a_rows")),
                                       L.ptr(_i32c(m_count, "m_count")), m_cap, L.ptr(wptr), cin, cout, L.ptr(_f32c(scale, "scale")),
                                       L.ptr(_f32c(shift, "shift")), relu, L.ptr(_i32c(relu_if_neg, "relu_if_neg")),
                                       L.ptr(_i32c(out_rows, "out_rows")), L.ptr(_f32rows(residual2d, "residual")),
                                       residual2d.stride(0) if residual2d is not None else 0, L.ptr(_f32rows(out2d, "out")),
                                       out2d.stride(0), L.ptr(_f32c(pool, "pool")), S, Sx, ho, wo, 1 if mode == "bf16x3" else 0,
                                       L.stream_ptr(out2d)), "ldn_conv_rows_pool")
        return out2d
    if dense_ok:
        hi, wi, ho, wo, stride = geom if geom is not None else (0, 0, 0, 0, 1)
        fn, wptr = ((lib.ldn_conv_rows_split, split_rows_weight(w)) if mode == "bf16x3" else (lib.ldn_conv_rows_f32, _f32c(w, "w")))
        L.check(fn(L.ptr(_f32rows(a2d, "a")), a2d.stride(0), L.ptr(_i32c(a_rows, "a_rows")), taps,
                                        L.ptr(_i32c(m_count, "m_count")), m_cap, L.ptr(wptr), cin, cout,
                                        L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), relu,
                                        L.ptr(_i32c(relu_if_neg, "relu_if_neg")), L.ptr(_i32c(out_rows, "out_rows")),
                                        L.ptr(_f32rows(residual2d, "residual")), residual2d.stride(0) if residual2d is not None else 0,
                                        L.ptr(_f32rows(out2d, "out")), out2d.stride(0), L.ptr(_f32c(post_sub, "post_sub")),
                                        L.ptr(_f32c(chan_mask, "chan_mask")), rows_per_image, classes, L.ptr(_i32c(pix_map, "pix_map")),
                                        hi, wi, ho, wo, stride, L.ptr(_f32c(ln_stats, "ln_stats")), L.ptr(_f32c(ln_c1, "ln_c1")),
                                        L.stream_ptr(out2d)), "ldn_conv_rows_split" if mode == "bf16x3" else "ldn_conv_rows_f32")
        return out2d
    L.check(lib.ldn_conv_rows(L.ptr(_f32c(a2d, "a")), a2d.stride(0), L.ptr(_i32c(a_rows, "a_rows")), taps,
                              L.ptr(_i32c(m_count, "m_count")), m_cap, L.ptr(_f32c(w, "w")), cin, cout,
                              L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), relu,
                              L.ptr(_i32c(relu_if_neg, "relu_if_neg")), L.ptr(_i32c(out_rows, "out_rows")),
                              L.ptr(_f32rows(residual2d, "residual")), residual2d.stride(0) if residual2d is not None else 0,
                              L.ptr(_f32rows(out2d, "out")), out2d.stride(0), _mm(math), L.stream_ptr(out2d)), "ldn_conv_rows")
    return out2d


USE_ROWS_PS = os.environ.get("LDN_ROWS_PS", "1") != "0"      # the pre-split packed path (k_dense<PS / OF> + k_rows3); 0 = round 4's three launches


ROWS_PS_MAX_WIDTH = 2048     # = ROWS3_MAX_CIN of csrc/ldn_rows3.hip


def rows_ps_ok(cin, width, cout):
    """Can a spatial / layer block keep h1 / h2 pre-split between its launches (ldn_conv_rows_ps + ldn_conv3x3_rows_ps)?  bf16x3 mode on
    the k_dense path, widths the kernels tile (conv1: cin % 32, width % 64; 3x3: width % 64 and width <= 2048 -- ldn_conv3x3_rows_ps reads a
    missing neighbour from a zero row of that width; conv3: cout % 64).  Wider blocks take conv_rows(taps=9)."""
    return (USE_ROWS_PS and USE_DENSE_KERNEL and get_math_mode() == "bf16x3" and cin % 32 == 0 and width % 64 == 0 and width <= ROWS_PS_MAX_WIDTH
            and cout % 64 == 0)


def conv_rows_ps(a2d, w, scale, shift, out2d, *, a_presplit=False, out_presplit=False, a_rows=None, m_count=None, m_cap=None, relu=1,
                 relu_if_neg=None, out_rows=None, residual2d=None, rows_hint=None, pool=None, pool_grid=None):
    """1x1 packed-row convolution with pre-split rows on the input and / or output side (see ldn_conv_rows_ps).  a2d / out2d are float32
    tensors either way (a pre-split row is the same 4 bytes per element); w [cout, 1, cin] fp32 (its pre-split copy is cached)."""
    L.require_device(a2d, w, out2d)
    lib = L.load()
    cout, t, cin = w.shape
    if t != 1:
        raise L.LdnError("conv_rows_ps: 1x1 weights expected")
    if m_cap is None:
        m_cap = a2d.shape[0] if a_rows is None else a_rows.numel()
    if rows_hint is not None and m_count is not None and USE_ROWS_HINT:
        lib.ldn_hint_rows(int(rows_hint))
    S, Sx, ho, wo = pool_grid if pool is not None else (0, 0, 0, 0)
    L.check(lib.ldn_conv_rows_ps(L.ptr(_f32rows(a2d, "a")), a2d.stride(0), 1 if a_presplit else 0, L.ptr(_i32c(a_rows, "a_rows")),
                                 L.ptr(_i32c(m_count, "m_count")), m_cap, L.ptr(split_rows_weight(w)), cin, cout,
                                 L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), relu, L.ptr(_i32c(relu_if_neg, "relu_if_neg")),
                                 L.ptr(_i32c(out_rows, "out_rows")), L.ptr(_f32rows(residual2d, "residual")),
                                 residual2d.stride(0) if residual2d is not None else 0, L.ptr(_f32rows(out2d, "out")), out2d.stride(0),
                                 1 if out_presplit else 0, L.ptr(_f32c(pool, "pool")), S, Sx, ho, wo, L.stream_ptr(out2d)), "ldn_conv_rows_ps")
    return out2d


def conv3x3_rows_ps(a2d, nbr, w, scale, shift, out2d, *, m_count=None, m_cap=None, relu=1, out_presplit=False, rows_hint=None):
    """The packed 3x3 on pre-split rows (k_rows3, see ldn_conv3x3_rows_ps): a2d [rows, cin] pre-split h1, nbr [m_cap, 9], w [cout, 9, cin]."""
    L.require_device(a2d, nbr, w, out2d)
    lib = L.load()
    cout, t, cin = w.shape
    if t != 9:
        raise L.LdnError("conv3x3_rows_ps: 3x3 weights [cout, 9, cin] expected")
    if m_cap is None:
        m_cap = nbr.numel() // 9
    hint = int(rows_hint) if (rows_hint is not None and USE_ROWS_HINT) else -1
    L.check(lib.ldn_conv3x3_rows_ps(L.ptr(_f32rows(a2d, "a")), a2d.stride(0), L.ptr(_i32c(nbr, "nbr")), L.ptr(_i32c(m_count, "m_count")), m_cap,
                                    L.ptr(split_rows_weight(w)), cin, cout, L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), relu,
                                    L.ptr(_f32rows(out2d, "out")), out2d.stride(0), 1 if out_presplit else 0, hint, L.stream_ptr(out2d)),
            "ldn_conv3x3_rows_ps")
    return out2d


def presplit_rows(x2d):
    """fp32 rows -> pre-split rows ([row][C / 8][8 hi | 8 lo] bf16 viewed as float32 [rows, C]); host-side helper for tests and tools
    (the kernels produce this layout themselves)."""
    rows, C = x2d.shape
    hi = x2d.to(torch.bfloat16)
    lo = (x2d - hi.float()).to(torch.bfloat16)
    packed = torch.stack((hi.view(rows, C // 8, 8), lo.view(rows, C // 8, 8)), dim=2).contiguous()     # [rows, C/8, 2, 8] bf16
    return packed.view(torch.float32).reshape(rows, C)


def unsplit_rows(ps2d):
    """Inverse of presplit_rows (hi + lo in fp32)."""
    rows, C = ps2d.shape
    b = ps2d.contiguous().view(torch.bfloat16).reshape(rows, C // 8, 2, 8).float()
    return (b[:, :, 0] + b[:, :, 1]).reshape(rows, C)


# ---------------------------------------------------------------------------------------- a2
def channel_masker(x_nhwc, w1, b1, w2, b2, groups, gran, mask_in=None, want_logits=False, gap_partial=None, hw=None):
    """Masker_channel_MLP eval forward + active channel lists (see ldn_channel_masker).
    Returns (mask [B,G], ch_idx [B,G*gran] int32, ch_cnt [B] int32, logits [B,2G] or None)."""
    lib = L.load()
    src = x_nhwc if x_nhwc is not None else (mask_in if mask_in is not None else gap_partial)
    dev = src.device
    L.require_device(x_nhwc, mask_in, gap_partial)
    if gap_partial is not None:
        x_nhwc = None
    gap_splits = 0
    if mask_in is not None:
        B = mask_in.shape[0]
        HW = C = hidden = 0
        work = None
    elif gap_partial is not None:   # fused GAP: [B, splits, C] channel sums left by the previous conv's epilogue
        B, gap_splits, C = gap_partial.shape
        HW = hw
        hidden = 0 if w2 is None else w1.shape[0]
        work = None
        _f32c(gap_partial, "gap_partial")
    else:
        B, H, W, C = x_nhwc.shape
        HW = H * W
        hidden = 0 if w2 is None else w1.shape[0]
        work = _work(lib.ldn_channel_masker_workspace_bytes(B, HW, C), dev)
    width = groups * gran
    mask = torch.empty(B, groups, device=dev, dtype=torch.float32)
    idx = torch.empty(B, width, device=dev, dtype=torch.int32)
    cnt = torch.empty(B, device=dev, dtype=torch.int32)
    logits = torch.empty(B, 2 * groups, device=dev, dtype=torch.float32) if (want_logits and mask_in is None) else None
    L.check(lib.ldn_channel_masker(L.ptr(x_nhwc), B, HW, C, L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), hidden, groups,
                                   gran, L.ptr(_f32c(mask_in, "mask_in") if mask_in is not None else None), L.ptr(mask),
                                   L.ptr(logits), L.ptr(idx), L.ptr(cnt), L.ptr(work), L.ptr(gap_partial), gap_splits,
                                   L.stream_ptr()), "ldn_channel_masker")
    return mask, idx, cnt, logits


# ---------------------------------------------------------------------------------------- a7 image
def conv_image(a_nhwc, w, scale, shift, out_nhwc, *, ksize=1, stride=1, k_idx=None, k_cnt=None, kgran=1, n_idx=None,
               n_cnt=None, post_sub=None, relu=1, residual=None, colsum=None, math=None, out_split=False):
    """Per-image channel-subset convolution (see ldn_conv_image).
    a_nhwc [B,Hi,Wi,lda]; w [cout,ksize*ksize,cin] without k_idx, [ksize*ksize,cin,cout] (k-major) with k_idx;
    shift [cout] or [16,cout]; out_nhwc [B,Ho,Wo,ldo]."""
    L.require_device(a_nhwc, w, out_nhwc)
    lib = L.load()
    B, Hi, Wi, lda = a_nhwc.shape
    _, Ho, Wo, ldo = out_nhwc.shape
    if k_idx is None:
        cout, t, cin = w.shape
    else:
        t, cin, cout = w.shape
    if t != ksize * ksize:
        raise L.LdnError("conv_image: weight taps do not match ksize")
    classes = 1 if shift.dim() == 1 else shift.shape[0]
    L.check(lib.ldn_conv_image(L.ptr(_f32c(a_nhwc, "a")), lda, B, Hi, Wi, ksize, stride, Ho, Wo, L.ptr(_f32c(w, "w")),
                               cin, cout, L.ptr(_i32c(k_idx, "k_idx")), L.ptr(_i32c(k_cnt, "k_cnt")), kgran,
                               L.ptr(_i32c(n_idx, "n_idx")), L.ptr(_i32c(n_cnt, "n_cnt")),
                               L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), classes, L.ptr(post_sub),
                               relu, L.ptr(residual), residual.shape[-1] if residual is not None else 0,
                               L.ptr(_f32c(out_nhwc, "out")), ldo, L.ptr(colsum), 1 if out_split else 0, _mm(math),
                               L.stream_ptr(out_nhwc)),
            "ldn_conv_image")
    return out_nhwc


# ---------------------------------------------------------------------------------------- a7 packed (+ channel lists)
def conv_packed(a2d, w, scale, shift, out2d, *, B=1, row_prefix=None, m_count=None, m_cap=None, a_map=None, taps=1,
                out_map=None, pix_map=None, geom=None, k_idx=None, k_cnt=None, kgran=1, n_idx=None, n_cnt=None,
                post_sub=None, relu=1, relu_if_neg=None, residual2d=None, math=None):
    """Convolution over packed pixel lists with optional per-image channel subsets (see ldn_conv_packed).
    geom = (Hi, Wi, Ho, Wo, stride) is only needed with a 16-class shift table."""
    L.require_device(a2d, w, out2d)
    lib = L.load()
    if k_idx is None:
        cout, t, cin = w.shape
    else:
        t, cin, cout = w.shape
    if t != taps:
        raise L.LdnError(f"conv_packed: weight has {t} taps, expected {taps}")
    classes = 1 if shift.dim() == 1 else shift.shape[0]
    hi, wi, ho, wo, stride = geom if geom is not None else (0, 0, 0, 0, 1)
    L.check(lib.ldn_conv_packed(L.ptr(_f32c(a2d, "a")), a2d.stride(0), B, L.ptr(_i32c(row_prefix, "row_prefix")),
                                L.ptr(_i32c(m_count, "m_count")), m_cap, L.ptr(_i32c(a_map, "a_map")), taps,
                                L.ptr(_i32c(out_map, "out_map")), L.ptr(_i32c(pix_map, "pix_map")), hi, wi, ho, wo, stride,
                                L.ptr(_f32c(w, "w")), cin, cout, L.ptr(_i32c(k_idx, "k_idx")), L.ptr(_i32c(k_cnt, "k_cnt")),
                                kgran, L.ptr(_i32c(n_idx, "n_idx")), L.ptr(_i32c(n_cnt, "n_cnt")),
                                L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), classes, L.ptr(post_sub), relu,
                                L.ptr(_i32c(relu_if_neg, "relu_if_neg")), L.ptr(residual2d),
                                residual2d.stride(0) if residual2d is not None else 0, L.ptr(_f32rows(out2d, "out")),
                                out2d.stride(0), _mm(math), L.stream_ptr(out2d)), "ldn_conv_packed")
    return out2d


# ---------------------------------------------------------------------------------------- a9 (LAD-RegNet, layer skip)
def grouped_conv3x3_rows(a2d, nbr, w, group_width, scale, shift, out2d, *, m_count=None, m_cap=None, relu=1):
    """Grouped 3x3 conv + BN (+ReLU) over packed rows (see ldn_grouped_conv3x3_rows).  w [C,9,gw]."""
    L.require_device(a2d, nbr, w, out2d)
    lib = L.load()
    C = w.shape[0]
    m_cap = out2d.shape[0] if m_cap is None else m_cap
    L.check(lib.ldn_grouped_conv3x3_rows(L.ptr(_f32c(a2d, "a")), a2d.stride(0), L.ptr(_i32c(nbr, "nbr")),
                                         L.ptr(_i32c(m_count, "m_count")), m_cap, L.ptr(_f32c(w, "w")), C, group_width,
                                         L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), relu,
                                         L.ptr(_f32c(out2d, "out")), out2d.stride(0), L.stream_ptr()),
            "ldn_grouped_conv3x3_rows")
    return out2d


def pack_grouped16_weights(w):
    """w [C, 9, 16] fp32 (out channel, tap, in channel of the group) -> MFMA fragment order [C/16][5][64][hi 8 | lo 8] bf16
    (ldn_grouped16_conv3x3_rows): a K step is a pair of taps x 16 channels; lane = out channel i + 16 * k-group."""
    C = w.shape[0]
    if tuple(w.shape[1:]) != (9, 16) or C % 16:
        raise L.LdnError("pack_grouped16_weights: expected [C % 16 == 0, 9, 16]")
    k = torch.zeros(C, 10, 16, device=w.device, dtype=torch.float32)
    k[:, :9] = w.detach().float()
    k = k.reshape(C // 16, 16, 5, 2, 2, 8)                 # [g][i][s][tap in pair][channel half][e]: k-group kg = 2 * tap + half
    k = k.permute(0, 2, 3, 4, 1, 5).reshape(C // 16, 5, 64, 8)     # lane = i + 16 * kg
    hi, lo = _hi_lo(k)
    return torch.stack((hi, lo), dim=-2).contiguous()


def grouped16_conv3x3_rows(a2d, nbr, w_frag, scale, shift, out2d, *, m_count=None, m_cap=None, relu=1):
    """Grouped 3x3 conv (group width 16) + BN (+ReLU) over packed rows on the matrix cores (see ldn_grouped16_conv3x3_rows)."""
    L.require_device(a2d, nbr, w_frag, out2d)
    lib = L.load()
    C = w_frag.shape[0] * 16
    if w_frag.dtype != torch.bfloat16 or not w_frag.is_contiguous() or w_frag.numel() * 2 != lib.ldn_grouped16_weight_bytes(C):
        raise L.LdnError("grouped16_conv3x3_rows: w_frag must be the contiguous bf16 tensor of pack_grouped16_weights")
    m_cap = out2d.shape[0] if m_cap is None else m_cap
    L.check(lib.ldn_grouped16_conv3x3_rows(L.ptr(_f32c(a2d, "a")), a2d.stride(0), L.ptr(_i32c(nbr, "nbr")),
                                           L.ptr(_i32c(m_count, "m_count")), m_cap, L.ptr(w_frag), C, L.ptr(_f32c(scale, "scale")),
                                           L.ptr(_f32c(shift, "shift")), relu, L.ptr(_f32c(out2d, "out")), out2d.stride(0),
                                           L.stream_ptr()), "ldn_grouped16_conv3x3_rows")
    return out2d


def grouped16_images_fit(Hi, Wi, C):
    """Groups per workgroup of ldn_grouped16_conv3x3_images for an Hi x Wi input map (0: it does not fit the LDS)."""
    return int(L.load().ldn_grouped16_images_fit(int(Hi), int(Wi), int(C)))


def grouped16_conv3x3_images(a2d, w_frag, scale, shift, out2d, *, m_count, images, relu=1):
    """Grouped 3x3 conv (group width 16) + BN (+ReLU) over packed rows that are WHOLE IMAGES in order (see
    ldn_grouped16_conv3x3_images).  images = (B, Hi, Wi, Ho, Wo, stride); m_count = device-side number of output rows."""
    L.require_device(a2d, w_frag, out2d, m_count)
    lib = L.load()
    C = w_frag.shape[0] * 16
    if w_frag.dtype != torch.bfloat16 or not w_frag.is_contiguous() or w_frag.numel() * 2 != lib.ldn_grouped16_weight_bytes(C):
        raise L.LdnError("grouped16_conv3x3_images: w_frag must be the contiguous bf16 tensor of pack_grouped16_weights")
    B, Hi, Wi, Ho, Wo, stride = (int(v) for v in images)
    if a2d.shape[0] < B * Hi * Wi or out2d.shape[0] < B * Ho * Wo:
        raise L.LdnError("grouped16_conv3x3_images: a / out must hold B whole images")
    L.check(lib.ldn_grouped16_conv3x3_images(L.ptr(_f32c(a2d, "a")), a2d.stride(0), L.ptr(_i32c(m_count, "m_count")), B, Hi, Wi, Ho,
                                             Wo, stride, L.ptr(w_frag), C, L.ptr(_f32c(scale, "scale")),
                                             L.ptr(_f32c(shift, "shift")), relu, L.ptr(_f32c(out2d, "out")), out2d.stride(0),
                                             L.stream_ptr()), "ldn_grouped16_conv3x3_images")
    return out2d


def grouped16_images_bands(Hi, Wi, Ho, stride, C):
    """Bands of output rows per image of ldn_grouped16_conv3x3_images (1: a workgroup holds a whole image; 0: does not fit)."""
    return int(L.load().ldn_grouped16_images_bands(int(Hi), int(Wi), int(Ho), int(stride), int(C)))


def grouped16_conv3x3_images_gap(a2d, w_frag, scale, shift, out2d, *, m_count, images, relu=1):
    """grouped16_conv3x3_images that also returns the channel sums of its output per kept image and band: gap [B, bands, C]
    (ldn_grouped16_conv3x3_images_gap) -- the squeeze of the SE block that follows conv b, without a second pass over h_b."""
    L.require_device(a2d, w_frag, out2d, m_count)
    lib = L.load()
    C = w_frag.shape[0] * 16
    if w_frag.dtype != torch.bfloat16 or not w_frag.is_contiguous() or w_frag.numel() * 2 != lib.ldn_grouped16_weight_bytes(C):
        raise L.LdnError("grouped16_conv3x3_images_gap: w_frag must be the contiguous bf16 tensor of pack_grouped16_weights")
    B, Hi, Wi, Ho, Wo, stride = (int(v) for v in images)
    if a2d.shape[0] < B * Hi * Wi or out2d.shape[0] < B * Ho * Wo:
        raise L.LdnError("grouped16_conv3x3_images_gap: a / out must hold B whole images")
    bands = grouped16_images_bands(Hi, Wi, Ho, stride, C)
    if bands <= 0:
        raise L.LdnError("grouped16_conv3x3_images_gap: the map does not fit the LDS (grouped16_images_fit)")
    gap = torch.empty(B, bands, C, device=a2d.device, dtype=torch.float32)
    L.check(lib.ldn_grouped16_conv3x3_images_gap(L.ptr(_f32c(a2d, "a")), a2d.stride(0), L.ptr(_i32c(m_count, "m_count")), B, Hi, Wi, Ho,
                                                 Wo, stride, L.ptr(w_frag), C, L.ptr(_f32c(scale, "scale")),
                                                 L.ptr(_f32c(shift, "shift")), relu, L.ptr(_f32c(out2d, "out")), out2d.stride(0),
                                                 L.ptr(gap), L.stream_ptr()), "ldn_grouped16_conv3x3_images_gap")
    return gap


def se_gate_slots(gap, m_count, rows_per_image, w1, b1, w2, b2):
    """SE gate of every kept image from the channel sums of grouped16_conv3x3_images_gap (ldn_se_gate_slots) -> gate [B, C]."""
    L.require_device(gap, m_count, w1)
    lib = L.load()
    B, bands, C = gap.shape
    S = w1.shape[0]
    gate = torch.empty(B, C, device=gap.device, dtype=torch.float32)
    L.check(lib.ldn_se_gate_slots(L.ptr(_f32c(gap, "gap")), bands, L.ptr(_i32c(m_count, "m_count")), B, int(rows_per_image), C, S,
                                  L.ptr(_f32c(w1, "w1")), L.ptr(_f32c(b1, "b1")), L.ptr(_f32c(w2, "w2")), L.ptr(_f32c(b2, "b2")),
                                  L.ptr(gate), L.stream_ptr()), "ldn_se_gate_slots")
    return gate


def conv_rows_gated_fits(cin, gate_rows):
    """ldn_conv_rows_gated keeps the gate vectors of the images a 256-row tile touches in LDS (include/ldn_hip.h): does this shape fit?"""
    return cin % 8 == 0 and cin <= 2048 and (255 // int(gate_rows) + 2) * ((int(cin) + 31) // 32 * 32) * 4 <= 44 * 1024


def conv_rows_gated(a2d, w, scale, shift, out2d, gate, gate_rows, *, m_count=None, m_cap=None, relu=1, relu_if_neg=None, out_rows=None,
                    residual2d=None):
    """1x1 packed-row convolution whose input row r is multiplied by gate[r // gate_rows] in flight (ldn_conv_rows_gated; bf16x3)."""
    L.require_device(a2d, w, out2d, gate)
    lib = L.load()
    cout, t, cin = w.shape
    if t != 1:
        raise L.LdnError("conv_rows_gated: 1x1 weights expected")
    if gate.shape[1] != cin or not gate.is_contiguous():
        raise L.LdnError("conv_rows_gated: gate must be a contiguous [images, cin] tensor")
    if m_cap is None:
        m_cap = a2d.shape[0]
    if (m_cap + gate_rows - 1) // gate_rows > gate.shape[0]:
        raise L.LdnError("conv_rows_gated: gate has fewer rows than the images m_cap spans")
    L.check(lib.ldn_conv_rows_gated(L.ptr(_f32rows(a2d, "a")), a2d.stride(0), L.ptr(_i32c(m_count, "m_count")), m_cap,
                                    L.ptr(split_rows_weight(w)), cin, cout, L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), relu,
                                    L.ptr(_i32c(relu_if_neg, "relu_if_neg")), L.ptr(_i32c(out_rows, "out_rows")),
                                    L.ptr(_f32rows(residual2d, "residual")), residual2d.stride(0) if residual2d is not None else 0,
                                    L.ptr(_f32rows(out2d, "out")), out2d.stride(0), L.ptr(_f32c(gate, "gate")), int(gate_rows),
                                    L.stream_ptr(out2d)), "ldn_conv_rows_gated")
    return out2d


def grouped_conv3x3_image(a_nhwc, w, group_width, ch_idx, ch_cnt, scale, shift, out_nhwc, *, stride=1, relu=1):
    """Grouped 3x3 conv + BN (+ReLU) on left-packed per-image channel subsets (see ldn_grouped_conv3x3_image).  w [C,9,gw]."""
    L.require_device(a_nhwc, w, out_nhwc, ch_idx)
    lib = L.load()
    B, Hi, Wi, lda = a_nhwc.shape
    _, Ho, Wo, ldo = out_nhwc.shape
    C = w.shape[0]
    L.check(lib.ldn_grouped_conv3x3_image(L.ptr(_f32c(a_nhwc, "a")), lda, B, Hi, Wi, stride, Ho, Wo, L.ptr(_f32c(w, "w")), C,
                                          group_width, L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")),
                                          L.ptr(_f32c(scale, "scale")), L.ptr(_f32c(shift, "shift")), relu,
                                          L.ptr(_f32c(out_nhwc, "out")), ldo, L.stream_ptr(out_nhwc)), "ldn_grouped_conv3x3_image")
    return out_nhwc


def se_packed(a2d, row_prefix, w1, b1, w2, b2, max_rows_per_image, ch_idx=None, ch_cnt=None):
    """In-place squeeze-excitation over the packed rows of every kept image (see ldn_se_packed)."""
    L.require_device(a2d, row_prefix)
    lib = L.load()
    B = row_prefix.numel() - 1
    S, C = w1.shape
    work = _work(lib.ldn_se_packed_workspace_bytes(B, C, max_rows_per_image), a2d.device)
    L.check(lib.ldn_se_packed(L.ptr(_f32c(a2d, "a")), a2d.stride(0), L.ptr(_i32c(row_prefix, "row_prefix")), B, C, S,
                              L.ptr(_f32c(w1, "w1")), L.ptr(_f32c(b1, "b1")), L.ptr(_f32c(w2, "w2")), L.ptr(_f32c(b2, "b2")),
                              L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")), max_rows_per_image, L.ptr(work),
                              L.stream_ptr()), "ldn_se_packed")
    return a2d


# ---------------------------------------------------------------------------------------- a7 fused tail (channel mode, bf16x3)
def _hi_lo(w):
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi, lo


def pack_w2_pairs(conv2_weight, f32=False):
    """conv2.weight [W(out n), W(in k), 3, 3] fp32 -> the pair-interleaved, pre-split layout of ldn_bottleneck_tail:
    [9][W/2 kp][W/2 np][2 n][hi k0, hi k1, lo k0, lo k1] bf16 (one-time module preparation, include/ldn_hip.h).
    f32: the fp32 twin of ldn_bottleneck_tail_f32, [9][W/2 kp][W/2 np][2 n][w k0, w k1] fp32 -- the same 8 bytes per (n, k pair)."""
    W = conv2_weight.shape[0]
    w = conv2_weight.detach().float().permute(2, 3, 1, 0).reshape(9, W // 2, 2, W // 2, 2)   # [tap][kp][kk][np][nn]
    w = w.permute(0, 1, 3, 4, 2).contiguous()                                                 # [tap][kp][np][nn][kk]
    if f32:
        return w
    hi, lo = _hi_lo(w)
    return torch.stack((hi, lo), dim=-2).contiguous()                                         # [...][nn][hi/lo][kk]


def pack_w3_pairs(w3_scaled, f32=False):
    """bn3.scale * conv3.weight as [cout, W] fp32 -> [W/2 kp][cout][hi k0, hi k1, lo k0, lo k1] bf16 (f32: [W/2 kp][cout][w k0, w k1] fp32)."""
    cout, W = w3_scaled.shape
    w = w3_scaled.detach().float().t().reshape(W // 2, 2, cout).permute(0, 2, 1).contiguous()   # [kp][c][kk]
    if f32:
        return w
    hi, lo = _hi_lo(w)
    return torch.stack((hi, lo), dim=-2).contiguous()


def pack_w1_split(conv1_weight_2d, f32=False):
    """conv1.weight as [W, cin] fp32 -> [W][cin/8][hi 8 | lo 8] bf16 (n-major rows, pre-split; ldn_bottleneck_head).  f32: the plain
    contiguous fp32 matrix (ldn_bottleneck_head_f32: [n][octet][8 floats] IS row-major fp32)."""
    W, cin = conv1_weight_2d.shape
    if f32:
        return conv1_weight_2d.detach().float().contiguous()
    w = conv1_weight_2d.detach().float().reshape(W, cin // 8, 8)
    hi, lo = _hi_lo(w)
    return torch.stack((hi, lo), dim=-2).contiguous()


def bottleneck_head(x_nhwc, w1_split, ch_idx, ch_cnt, scale1, shift1, post_sub1, h1_split, x_split=None):
    """conv1 of a channel-mode block on the image's active output channels, written pre-split (see ldn_bottleneck_head).
    x_split (optional, x_split_buffer(B*H*W, cin)): receives x itself pre-split in 32-pixel tiles (ldn_bottleneck_head_split), the operand
    of a projection shortcut folded into the tail (bottleneck_tail_proj)."""
    L.require_device(x_nhwc, w1_split, h1_split)
    lib = L.load()
    B, H, Wd, cin = x_nhwc.shape
    width = ch_idx.shape[1]
    if w1_split.dtype == torch.float32:      # true-fp32 arithmetic (pack_w1_split(..., f32=True))
        if not w1_split.is_contiguous() or x_split is not None:
            raise L.LdnError("bottleneck_head: the fp32 form takes a contiguous fp32 weight matrix and writes no x_split")
        L.check(lib.ldn_bottleneck_head_f32(L.ptr(_f32c(x_nhwc, "x")), cin, B, H * Wd, cin, L.ptr(w1_split), width,
                                            L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")), L.ptr(_f32c(scale1, "scale1")),
                                            L.ptr(_f32c(shift1, "shift1")), L.ptr(_f32c(post_sub1, "post_sub1")),
                                            L.ptr(_f32c(h1_split, "h1")), h1_split.shape[-1], L.stream_ptr(h1_split)), "ldn_bottleneck_head_f32")
        return h1_split
    if w1_split.dtype != torch.bfloat16 or not w1_split.is_contiguous():
        raise L.LdnError("bottleneck_head: w1_split must be the contiguous bf16 tensor of pack_w1_split")
    if x_split is not None:
        if x_split.dtype != torch.float32 or not x_split.is_contiguous() or x_split.numel() * 4 < lib.ldn_x_split_bytes(B * H * Wd, cin):
            raise L.LdnError("bottleneck_head: x_split must be a contiguous fp32 buffer of x_split_buffer(B * H * W, cin)")
        L.check(lib.ldn_bottleneck_head_split(L.ptr(_f32c(x_nhwc, "x")), cin, B, H * Wd, cin, L.ptr(w1_split), width,
                                              L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")), L.ptr(_f32c(scale1, "scale1")),
                                              L.ptr(_f32c(shift1, "shift1")), L.ptr(_f32c(post_sub1, "post_sub1")),
                                              L.ptr(_f32c(h1_split, "h1")), h1_split.shape[-1], L.ptr(x_split),
                                              L.stream_ptr(h1_split)), "ldn_bottleneck_head_split")
        return h1_split
    L.check(lib.ldn_bottleneck_head(L.ptr(_f32c(x_nhwc, "x")), cin, B, H * Wd, cin, L.ptr(w1_split), width,
                                    L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")), L.ptr(_f32c(scale1, "scale1")),
                                    L.ptr(_f32c(shift1, "shift1")), L.ptr(_f32c(post_sub1, "post_sub1")),
                                    L.ptr(_f32c(h1_split, "h1")), h1_split.shape[-1], L.stream_ptr(h1_split)), "ldn_bottleneck_head")
    return h1_split


CHAIN_BLOCK_FIELDS = 14   # pointers of one ldn_chain_block


def chain_table(rows, device):
    """Device array of ldn_chain_block for ldn_bottleneck_chain.  rows: per block the 14 tensors (or None) in the order of the
    struct (w1_split, scale1, shift1, post_sub1, w2_pairs, w3_pairs, scale2, shift2_tab, post_sub2, shift3, mw1, mb1, mw2, mb2).
    The caller keeps the tensors alive for as long as it uses the table."""
    flat = []
    for row in rows:
        if len(row) != CHAIN_BLOCK_FIELDS:
            raise L.LdnError("chain_table: a block takes 14 tensors")
        for t in row:
            if t is not None and (not t.is_contiguous() or t.device != torch.device(device)):
                raise L.LdnError("chain_table: tensors must be contiguous and on the table's device")
            flat.append(0 if t is None else t.data_ptr())
    return torch.tensor(flat, dtype=torch.int64).view(len(rows), CHAIN_BLOCK_FIELDS).to(device)


def bottleneck_chain(x_in, x_work, table, width, hidden, G, gran, gap_in, f32=False):
    """A run of stride-1 channel-mode bottlenecks as one launch (see ldn_bottleneck_chain).  x_in / x_work [B,H,Wd,C] NHWC fp32
    (may be the same tensor); table = chain_table(...); gap_in [B,splits,C].
    Returns (masks [n,B,G], ch_idx [n,B,width], ch_cnt [n,B], colsum [B,8,C]); x_work holds the run's output."""
    L.require_device(x_in, x_work, table, gap_in)
    lib = L.load()
    B, H, Wd, C = x_in.shape
    n = table.shape[0]
    dev = x_in.device
    if x_work.shape != x_in.shape or not (x_in.is_contiguous() and x_work.is_contiguous()) or x_in.dtype != torch.float32:
        raise L.LdnError("bottleneck_chain: x_in / x_work must be contiguous fp32 NHWC tensors of the same shape")
    if table.dtype != torch.int64 or table.dim() != 2 or table.shape[1] != CHAIN_BLOCK_FIELDS or not table.is_contiguous():
        raise L.LdnError("bottleneck_chain: table must come from chain_table")
    if gap_in.dim() != 3 or gap_in.shape[0] != B or gap_in.shape[2] != C:
        raise L.LdnError("bottleneck_chain: gap_in must be [B, splits, C]")
    masks = torch.empty(n, B, G, device=dev, dtype=torch.float32)
    ch_idx = torch.empty(n, B, width, device=dev, dtype=torch.int32)
    ch_cnt = torch.empty(n, B, device=dev, dtype=torch.int32)
    colsum = torch.empty(B, 8, C, device=dev, dtype=torch.float32)
    h1 = torch.empty(B, H, Wd, width, device=dev, dtype=torch.float32)
    L.check((lib.ldn_bottleneck_chain_f32 if f32 else lib.ldn_bottleneck_chain)(L.ptr(x_in), L.ptr(x_work), C, B, H, Wd, C, width, L.ptr(table), n, hidden, G, gran,
                                     L.ptr(_f32c(gap_in, "gap_in")), gap_in.shape[1], L.ptr(colsum), L.ptr(masks), L.ptr(ch_idx),
                                     L.ptr(ch_cnt), L.ptr(h1), width, L.stream_ptr(x_work)), "ldn_bottleneck_chain")
    return masks, ch_idx, ch_cnt, colsum


def bottleneck_chain_fits(H, W, C, width, hidden, G):
    """Does a chained run on an H x W map fit the workgroup's LDS in every phase (ldn_bottleneck_chain_fits)?"""
    return bool(L.load().ldn_bottleneck_chain_fits(H, W, C, width, hidden, G))


def bottleneck_tail_splits(H, W, width, stride=1):
    """GAP partial slots of the fused tail whose 3x3 reads an H x W map with this stride (0: the map / width does not fit
    ldn_bottleneck_tail)."""
    return L.load().ldn_bottleneck_tail_splits(H, W, width, stride)


def bottleneck_tail(h1_split, w2_pairs, w3_pairs, ch_idx, ch_cnt, scale2, shift2_tab, post_sub2, shift3, out_nhwc, *,
                    residual=None, colsum=None, stride=1):
    """Fused conv2 -> conv3 tail of a channel-mode bottleneck (see ldn_bottleneck_tail).  h1_split [B,H,Wd,ldh] as written by
    bottleneck_head / conv_image(..., out_split=True); out_nhwc [B,Ho,Wo,cout] with Ho = (H-1)//stride+1 (stride = conv2's)."""
    L.require_device(h1_split, w2_pairs, w3_pairs, out_nhwc)
    lib = L.load()
    B, H, Wd, ldh = h1_split.shape
    width = ch_idx.shape[1]
    cout = out_nhwc.shape[-1]
    Ho, Wo = (H - 1) // stride + 1, (Wd - 1) // stride + 1
    if tuple(out_nhwc.shape) != (B, Ho, Wo, cout) or (residual is not None and tuple(residual.shape[:3]) != (B, Ho, Wo)):
        raise L.LdnError(f"bottleneck_tail: out / residual must be [B={B}, {Ho}, {Wo}, cout] for an {H}x{Wd} input at stride {stride}")
    if w2_pairs.dtype != w3_pairs.dtype or w2_pairs.dtype not in (torch.bfloat16, torch.float32) or not (w2_pairs.is_contiguous() and w3_pairs.is_contiguous()):
        raise L.LdnError("bottleneck_tail: w2_pairs / w3_pairs must be the contiguous tensors of pack_w2_pairs / pack_w3_pairs (both bf16, or both fp32)")
    fn = lib.ldn_bottleneck_tail if w2_pairs.dtype == torch.bfloat16 else lib.ldn_bottleneck_tail_f32
    L.check(fn(L.ptr(_f32c(h1_split, "h1")), ldh, B, H, Wd, stride, width, L.ptr(w2_pairs), L.ptr(w3_pairs), cout,
                                    L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")), L.ptr(_f32c(scale2, "scale2")),
                                    L.ptr(_f32c(shift2_tab, "shift2_tab")), L.ptr(_f32c(post_sub2, "post_sub2")),
                                    L.ptr(_f32c(shift3, "shift3")), L.ptr(residual),
                                    residual.shape[-1] if residual is not None else 0, L.ptr(_f32c(out_nhwc, "out")), cout,
                                    L.ptr(colsum), L.stream_ptr(out_nhwc)), "ldn_bottleneck_tail")
    return out_nhwc


def x_split_buffer(pixels, cin, dev):
    """Storage of a pre-split activation tensor in 32-pixel tiles (ldn_x_split_bytes; see ldn_bottleneck_head_split)."""
    return torch.empty(L.load().ldn_x_split_bytes(pixels, cin) // 4, device=dev, dtype=torch.float32)


def decode_x_split(xs, pixels, cin):
    """x_split buffer -> [pixels, cin] fp32 (hi + lo); test / debug helper."""
    t = xs.view(torch.bfloat16).reshape(-1, cin // 16, 2, 2, 32, 8).float()       # [tile][s][h][hi | lo][pixel][8]
    v = (t[:, :, :, 0] + t[:, :, :, 1]).permute(0, 3, 1, 2, 4).reshape(-1, cin)     # [tile * 32][s, h, 8] = channel 16 s + 8 h + i
    return v[:pixels]


def bottleneck_tail_proj_fits(H, W, width, cin):
    """Can the fused tail fold this block's stride-1 projection shortcut into conv3 (ldn_bottleneck_tail_proj_fits)?"""
    return bool(L.load().ldn_bottleneck_tail_proj_fits(H, W, width, cin))


def bottleneck_tail_proj(h1_split, w2_pairs, w3_pairs, ch_idx, ch_cnt, scale2, shift2_tab, post_sub2, shift3d, x_split, wd_pairs,
                         out_nhwc, *, colsum=None):
    """The fused tail with the projection shortcut as 64 more K values of conv3 (see ldn_bottleneck_tail_proj): x_split from
    bottleneck_head(..., x_split=...), wd_pairs = pack_w3_pairs(bn_d.scale * downsample.weight), shift3d = shift3 + bn_d.shift."""
    L.require_device(h1_split, w2_pairs, w3_pairs, x_split, wd_pairs, out_nhwc)
    lib = L.load()
    B, H, Wd, ldh = h1_split.shape
    width = ch_idx.shape[1]
    cout = out_nhwc.shape[-1]
    cin = wd_pairs.shape[0] * 2
    if tuple(out_nhwc.shape) != (B, H, Wd, cout) or x_split.numel() * 4 < lib.ldn_x_split_bytes(B * H * Wd, cin) or not x_split.is_contiguous():
        raise L.LdnError(f"bottleneck_tail_proj: out must be [B={B}, {H}, {Wd}, cout], x_split the buffer bottleneck_head filled")
    for t in (w2_pairs, w3_pairs, wd_pairs):
        if t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise L.LdnError("bottleneck_tail_proj: w2_pairs / w3_pairs / wd_pairs must be the contiguous bf16 tensors of pack_w2_pairs / pack_w3_pairs")
    L.check(lib.ldn_bottleneck_tail_proj(L.ptr(_f32c(h1_split, "h1")), ldh, B, H, Wd, width, L.ptr(w2_pairs), L.ptr(w3_pairs), cout,
                                         L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")), L.ptr(_f32c(scale2, "scale2")),
                                         L.ptr(_f32c(shift2_tab, "shift2_tab")), L.ptr(_f32c(post_sub2, "post_sub2")),
                                         L.ptr(_f32c(shift3d, "shift3d")), L.ptr(x_split), cin, L.ptr(wd_pairs),
                                         L.ptr(_f32c(out_nhwc, "out")), cout, L.ptr(colsum), L.stream_ptr(out_nhwc)), "ldn_bottleneck_tail_proj")
    return out_nhwc


def bottleneck_smallmap_fits(H, W, cin, width, cout):
    """Does a whole bottleneck on an H x W map fit one workgroup (ldn_bottleneck_smallmap_fits)?"""
    return bool(L.load().ldn_bottleneck_smallmap_fits(H, W, cin, width, cout))


def bottleneck_smallmap(x_nhwc, w1_split, w2_pairs, w3_pairs, ch_idx, ch_cnt, scale1, shift1, post_sub1, scale2, shift2_tab, post_sub2,
                        shift3, out_nhwc, *, residual=None, colsum=None):
    """A whole stride-1 channel-mode bottleneck on a small map as one launch (see ldn_bottleneck_smallmap).  x_nhwc [B,H,Wd,cin],
    out_nhwc / residual [B,H,Wd,cout] (may be the same tensor as x_nhwc); colsum [B,2,cout]."""
    L.require_device(x_nhwc, w1_split, w2_pairs, w3_pairs, out_nhwc)
    lib = L.load()
    B, H, Wd, cin = x_nhwc.shape
    width = ch_idx.shape[1]
    cout = out_nhwc.shape[-1]
    if tuple(out_nhwc.shape) != (B, H, Wd, cout) or (residual is not None and tuple(residual.shape) != (B, H, Wd, cout)):
        raise L.LdnError(f"bottleneck_smallmap: out / residual must be [B={B}, {H}, {Wd}, cout]")
    if colsum is not None and (tuple(colsum.shape) != (B, 2, cout) or not colsum.is_contiguous()):
        raise L.LdnError("bottleneck_smallmap: colsum must be a contiguous [B, 2, cout] tensor")
    for w in (w1_split, w2_pairs, w3_pairs):
        if w.dtype != torch.bfloat16 or not w.is_contiguous():
            raise L.LdnError("bottleneck_smallmap: weights must be the contiguous bf16 tensors of pack_w1_split / pack_w2_pairs / pack_w3_pairs")
    L.check(lib.ldn_bottleneck_smallmap(L.ptr(_f32c(x_nhwc, "x")), cin, B, H, Wd, cin, width, L.ptr(w1_split), L.ptr(w2_pairs),
                                        L.ptr(w3_pairs), cout, L.ptr(_i32c(ch_idx, "ch_idx")), L.ptr(_i32c(ch_cnt, "ch_cnt")),
                                        L.ptr(_f32c(scale1, "scale1")), L.ptr(_f32c(shift1, "shift1")), L.ptr(_f32c(post_sub1, "post_sub1")),
                                        L.ptr(_f32c(scale2, "scale2")), L.ptr(_f32c(shift2_tab, "shift2_tab")),
                                        L.ptr(_f32c(post_sub2, "post_sub2")), L.ptr(_f32c(shift3, "shift3")), L.ptr(residual),
                                        cout if residual is not None else 0, L.ptr(_f32c(out_nhwc, "out")), cout, L.ptr(colsum),
                                        L.stream_ptr(out_nhwc)), "ldn_bottleneck_smallmap")
    return out_nhwc


# ---------------------------------------------------------------------------------------- a8: the static stem
def pack_stem_weights(w_scaled):
    """bn1-scaled conv1.weight [cout,3,7,7] fp32 -> MFMA fragment order [cout/32][11][64][hi 8 | lo 8] bf16 (ldn_stem_conv_pool):
    K is 7 kernel rows x 24 slots (slot i < 21 = (kx = i // 3, c = i % 3), 3 zero pad slots), padded to 11 steps of 16."""
    cout = w_scaled.shape[0]
    if tuple(w_scaled.shape[1:]) != (3, 7, 7) or cout % 32:
        raise L.LdnError("pack_stem_weights: expected [cout % 32 == 0, 3, 7, 7]")
    w = w_scaled.detach().float().permute(0, 2, 3, 1).reshape(cout, 7, 21)         # [n][ky][kx*3 + c]
    k = torch.zeros(cout, 8, 24, device=w.device, dtype=torch.float32)             # + zero row ky == 7 (step 10, upper half)
    k[:, :7, :21] = w
    k = k.reshape(cout, 24, 8)[:, :22]                                             # k-groups q = 3 ky + g; q == 21 is the zero pad
    k = k.reshape(cout // 32, 32, 11, 2, 8)                                        # [j][n][s][h][e]: q = 2 s + h
    k = k.permute(0, 2, 3, 1, 4).reshape(cout // 32, 11, 64, 8)                    # lane = h * 32 + n
    hi, lo = _hi_lo(k)
    return torch.stack((hi, lo), dim=-2).contiguous()                              # [j][s][lane][2][8]


def stem_conv_pool(x_nhwc, w_frag, shift, cout, want_gap=False):
    """relu(maxpool3x3s2p1(conv7x7s2p3(x, w)) + shift) in one launch (see ldn_stem_conv_pool).  x_nhwc [B,H,W,3] -> [B,Hp,Wp,cout]
    (want_gap: -> (out, gap [B, splits, cout]) with the per-tile channel sums of out, ldn_stem_conv_pool_gap)."""
    L.require_device(x_nhwc, w_frag, shift)
    lib = L.load()
    B, H, W, cin = x_nhwc.shape
    if cin != 3 or x_nhwc.dtype != torch.float32 or not x_nhwc.is_contiguous():
        raise L.LdnError("stem_conv_pool: x must be a contiguous fp32 [B,H,W,3] tensor")
    if w_frag.dtype != torch.bfloat16 or not w_frag.is_contiguous() or w_frag.numel() * 2 != lib.ldn_stem_weight_bytes(cout):
        raise L.LdnError("stem_conv_pool: w_frag must be the contiguous bf16 tensor of pack_stem_weights")
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty(B, Hp, Wp, cout, device=x_nhwc.device, dtype=torch.float32)
    if want_gap:
        gap = torch.empty(B, lib.ldn_stem_gap_splits(H, W), cout, device=x_nhwc.device, dtype=torch.float32)
        L.check(lib.ldn_stem_conv_pool_gap(L.ptr(x_nhwc), B, H, W, L.ptr(w_frag), L.ptr(_f32c(shift, "shift")), cout, L.ptr(out), Hp, Wp,
                                           L.ptr(gap), L.stream_ptr(out)), "ldn_stem_conv_pool_gap")
        return out, gap
    L.check(lib.ldn_stem_conv_pool(L.ptr(x_nhwc), B, H, W, L.ptr(w_frag), L.ptr(_f32c(shift, "shift")), cout, L.ptr(out), Hp, Wp,
                                   L.stream_ptr(out)), "ldn_stem_conv_pool")
    return out


def pack_stem3_weights(w_scaled):
    """BN-scaled stem conv weight [cout,3,3,3] fp32 -> MFMA fragment order [cout/32][3][64][hi 8 | lo 8] bf16 (ldn_stem3_conv):
    K is 3 kernel rows x 16 slots (slot i < 9 = (kx = i // 3, c = i % 3), 7 zero pad slots), one K16 step per kernel row."""
    cout = w_scaled.shape[0]
    if tuple(w_scaled.shape[1:]) != (3, 3, 3) or cout % 32:
        raise L.LdnError("pack_stem3_weights: expected [cout % 32 == 0, 3, 3, 3]")
    k = torch.zeros(cout, 3, 16, device=w_scaled.device, dtype=torch.float32)
    k[:, :, :9] = w_scaled.detach().float().permute(0, 2, 3, 1).reshape(cout, 3, 9)   # [n][ky][kx*3 + c]
    k = k.reshape(cout // 32, 32, 3, 2, 8).permute(0, 2, 3, 1, 4).reshape(cout // 32, 3, 64, 8)   # lane = h * 32 + n
    hi, lo = _hi_lo(k)
    return torch.stack((hi, lo), dim=-2).contiguous()                                  # [j][s][lane][2][8]


def stem3_conv(x_nhwc, w_frag, shift, cout, relu=True):
    """act(conv3x3s2p1(x, w) + shift) in one launch (see ldn_stem3_conv).  x_nhwc [B,H,W,3] -> [B,Ho,Wo,cout]."""
    L.require_device(x_nhwc, w_frag, shift)
    lib = L.load()
    B, H, W, cin = x_nhwc.shape
    if cin != 3 or x_nhwc.dtype != torch.float32 or not x_nhwc.is_contiguous():
        raise L.LdnError("stem3_conv: x must be a contiguous fp32 [B,H,W,3] tensor")
    if w_frag.dtype != torch.bfloat16 or not w_frag.is_contiguous() or w_frag.numel() * 2 != lib.ldn_stem3_weight_bytes(cout):
        raise L.LdnError("stem3_conv: w_frag must be the contiguous bf16 tensor of pack_stem3_weights")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty(B, Ho, Wo, cout, device=x_nhwc.device, dtype=torch.float32)
    L.check(lib.ldn_stem3_conv(L.ptr(x_nhwc), B, H, W, L.ptr(w_frag), L.ptr(_f32c(shift, "shift")), cout, int(bool(relu)),
                               L.ptr(out), Ho, Wo, L.stream_ptr(out)), "ldn_stem3_conv")
    return out


# ---------------------------------------------------------------------------------------- a14: token skipping
def token_lists(keep):
    """keep [B, L] float {0,1} -> (tok_rows [B*L] int32: flat rows of the kept tokens, first `count` valid; prefix [B+1] int32;
    count [1] int32 on the device).  The index kernel of the pixel masks does it: a [B, L, 1] patch mask on an L x 1 map."""
    B, Lt = keep.shape
    ix = mask_to_index(keep.reshape(B, Lt, 1).float().contiguous(), Lt, 1, 1)
    return ix.idx3, ix.pre3, ix.cnt[0:1]


def packed_mha(qkv2d, tok_rows, prefix, B, heads, max_tokens, scale=None, head_keep=None):
    """Multi-head attention among the kept tokens of every image (see ldn_packed_mha).  qkv2d [B*L, 3*dim]; returns the packed
    rows [B*L (capacity), dim]: row n belongs to token tok_rows[n].  head_keep [B, heads] {0,1}: head skipping (ldn_packed_mha_heads)."""
    L.require_device(qkv2d, tok_rows, prefix)
    lib = L.load()
    rows, three_dim = qkv2d.shape
    dim = three_dim // 3
    d = dim // heads
    if three_dim != 3 * dim or dim != heads * d or qkv2d.dtype != torch.float32 or qkv2d.stride(1) != 1:
        raise L.LdnError("packed_mha: qkv must be fp32 [rows, 3 * heads * head_dim]")
    out = torch.empty(rows, dim, device=qkv2d.device, dtype=torch.float32)
    if head_keep is not None:
        if tuple(head_keep.shape) != (B, heads):
            raise L.LdnError("packed_mha: head_keep must be [B, heads]")
        L.check(lib.ldn_packed_mha_heads(L.ptr(qkv2d), qkv2d.stride(0), L.ptr(_i32c(tok_rows, "tok_rows")), L.ptr(_i32c(prefix, "prefix")), B,
                                         heads, d, max_tokens, float(scale if scale is not None else d ** -0.5),
                                         L.ptr(_f32c(head_keep, "head_keep")), L.ptr(out), dim, L.stream_ptr(out)), "ldn_packed_mha_heads")
        return out
    L.check(lib.ldn_packed_mha(L.ptr(qkv2d), qkv2d.stride(0), L.ptr(_i32c(tok_rows, "tok_rows")), L.ptr(_i32c(prefix, "prefix")), B, heads,
                               d, max_tokens, float(scale if scale is not None else d ** -0.5), L.ptr(out), dim, L.stream_ptr(out)),
            "ldn_packed_mha")
    return out
