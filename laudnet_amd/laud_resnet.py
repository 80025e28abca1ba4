"""LAUD-ResNet on the MI355X HIP path: host-side mirror of the reference's nn.Module surface.

Same class names, constructor kwargs, sub-module attribute names (hence state_dict keys/shapes),
`forward(x, temperature)` 7-tuple and `get_optim_policies()` as the reference's
imagenet_classification/models/laud_resnet.py + models/utils.py, so its checkpoints load unchanged
(SURVEY.md 8b).  What differs is how a block is executed: instead of dense conv x mask emulation
(laud_resnet.py:115-133) the dynamic bottleneck runs only the active work through libldn_hip.so:

  spatial / layer : patch mask -> ldn_mask_to_index -> conv1 on the dilated pixel list (gather on load)
                    -> 3x3 conv through the neighbour table -> 1x1 conv + BN + residual scatter-add + ReLU
  channel         : GAP+MLP masker -> per-image active-channel lists -> three ragged per-image GEMMs
                    (output-, input/output-, input-channel subsets), pre-BN mask constants folded into
                    image-independent shift tables (DESIGN.md "channel algebra")

The hot path has NO PyTorch/CPU fallback: training mode, CPU tensors or a missing library raise.
Build-only extension (default off): `forced_spatial_mask` / `forced_channel_mask` on a block inject
masks for "identical inputs/masks" parity runs (mask_override).
"""
from __future__ import annotations

import itertools
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import LdnError


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

__all__ = ["uni_resnet50", "uni_resnet101", "ResNet", "Bottleneck", "Masker_spatial", "Masker_channel_MLP",
           "Masker_channel_conv_linear", "ExpandMask"]


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, 3, stride=stride, padding=dilation, groups=groups, bias=False,
                     dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, 1, stride=stride, bias=bias)


def _fold_bn(bn: nn.BatchNorm2d):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    return scale.float().contiguous(), shift.float().contiguous()


def _eval_only(module: nn.Module, x: torch.Tensor):
    if module.training:
        raise LdnError("laudnet_amd.laud_resnet implements the eval-mode (inference) hot path; call model.eval().  The part of training that "
                       "is sparse -- spatial / layer blocks under frozen BatchNorm -- is laudnet_amd.training.sparse_block_train")
    if not x.is_cuda:
        raise LdnError("laudnet_amd has no CPU path: move the model and input to a HIP device")


_SIDE_STREAMS = {}
# Projection shortcuts on a side stream next to conv1 / conv2?  Round 1: yes (the main chain's launches left CUs idle).  With the
# one-workgroup-per-CU kernels of round 2 the overlap only stole their CUs (inline 1.3 % faster: 16.05 vs 16.27 ms).  Round 4: the
# first blocks of stages 2-4 are k_head + one strided tail launch whose conv2 phase leaves the memory pipe idle, and the projection
# (k_dense) fills it: 12.16-12.19 -> 12.09 ms, four interleaved runs on one box.  LDN_SIDE_STREAM=0 runs it inline.
_USE_SIDE_STREAM = __import__("os").environ.get("LDN_SIDE_STREAM", "1") == "1"


def _side_stream(dev):
    """One auxiliary stream per device for work that is independent of the main chain (projection shortcuts)."""
    key = str(dev)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def _after_load(module, _incompatible_keys):
    module._drop_cache()


_PREP_GEN = itertools.count()


class _PrepCache(nn.Module):
    """Mixin: lazily built, device-resident folded weights.  The cache is keyed on (data_ptr, _version) of every parameter
    and buffer of the module's OWN sub-tree, so in-place edits through the tensor itself (p.copy_ / p.mul_ under no_grad, BN
    re-estimation, an optimizer step) are noticed like load_state_dict / .to() / train() are.  NOT noticed: edits through
    `p.data` (`p.data.copy_(...)` -- `.data` is a detached alias with its own version counter, p._version stays put): call
    `invalidate()` on the module after such an edit, or the stale folded weights keep being used."""

    def _init_cache(self):
        self._prep = None
        self._prep_key = None
        self._prep_src = None
        self.register_load_state_dict_post_hook(_after_load)   # a module-level function: the module stays picklable

    def _drop_cache(self):
        self._prep = None
        self._prep_key = None
        self._prep_src = None

    invalidate = _drop_cache

    def _src_key(self):
        src = self._prep_src
        if src is None:
            src = self._prep_src = [t for t in list(self.parameters()) + list(self.buffers())]
        return tuple((t.data_ptr(), t._version) for t in src)

    def _cache_valid(self):
        """True when self._prep was built from the tensors as they are now."""
        return self._prep is not None and self._prep_key == self._src_key()

    def _cache_store(self, prep):
        self._prep = prep
        self._prep_key = self._src_key()
        self._prep_gen = next(_PREP_GEN)      # identifies THIS build of the cache (tables of device pointers are keyed on it)
        return prep

    def train(self, mode: bool = True):
        self._drop_cache()
        return super().train(mode)

    def _apply(self, fn, *a, **kw):
        self._drop_cache()
        return super()._apply(fn, *a, **kw)


# ------------------------------------------------------------------------------------------- maskers
class Masker_spatial(_PrepCache):
    """models/utils.py:35-65.  HIP: ldn_spatial_masker."""

    def __init__(self, in_channels, mask_channel_group, mask_size):
        super().__init__()
        self.mask_channel_group = mask_channel_group
        self.mask_size = mask_size
        self.conv = conv1x1(in_channels, mask_channel_group * 2, bias=True)
        self.conv_flops_pp = self.conv.weight.shape[0] * self.conv.weight.shape[1] + self.conv.weight.shape[1]
        self.conv.bias.data[:mask_channel_group] = 5.0
        self.conv.bias.data[mask_channel_group + 1:] = 0.0
        self._init_cache()

    def flops_for(self, x):
        if self.mask_size < x.shape[2]:
            h = w = self.mask_size
        else:
            h, w = x.shape[2], x.shape[3]
        return x.shape[1] * h * w + self.conv_flops_pp * h * w

    def forward(self, x, temperature, want_logits=False, carry=None):
        """carry = (work, prefix) of the previous layer-skip block on the same residual stream (ldn_spatial_masker): the images that
        block skipped are unchanged, their channel sums are reused.  self.last_work = this call's sums (None unless mask_size 1)."""
        _eval_only(self, x)
        if not self._cache_valid():
            with torch.no_grad():
                self._cache_store((self.conv.weight.detach().reshape(self.conv.weight.shape[0], -1).float().contiguous(),
                                   self.conv.bias.detach().float().contiguous()))
        w, b = self._prep
        mask, logits, self.last_work = ops.spatial_masker(ops.as_nhwc(x), w, b, self.mask_channel_group, self.mask_size, want_logits,
                                                          carry=carry, return_work=True)
        out = (mask, mask.mean(), self.flops_for(x))
        return out + (logits,) if want_logits else out

    def _wb(self):
        if not self._cache_valid():
            with torch.no_grad():
                self._cache_store((self.conv.weight.detach().reshape(self.conv.weight.shape[0], -1).float().contiguous(),
                                   self.conv.bias.detach().float().contiguous()))
        return self._prep

    def decide_from_means(self, work, x_shape, out_hw, patch_major, stride=1):
        """The decision AND the packed lists of the block in one launch, from the pooled patch means the previous block's conv3
        left in `work` (ldn_mask_plan; models/utils.py:47-65 without reading x).  Returns (mask [B,1,S,S], IndexSet)."""
        B, C, H, W = x_shape
        S = self.mask_size
        w, b = self._wb()
        mask, _, ix = ops.mask_plan(work.view(B, S, S, C), w, b, out_hw[0], out_hw[1], stride, patch_major=patch_major)
        self.last_work = work
        return mask, ix

    def decide_layer_pooled(self, x, S):
        """Layer skip at the start of a fused run: the channel means of the S x S tiles of every image (one pass over x, as the
        global pool is) stay in last_work for the blocks behind; the decision is taken from them (ldn_layer_head)."""
        w, b = self._wb()
        _, _, self.last_work = ops.spatial_masker(ops.as_nhwc(x), w, b, self.mask_channel_group, S, False, return_work=True)
        B, C = x.shape[0], x.shape[1]
        return ops.layer_head(self.last_work.view(B, S * S, C), w, b, self.mask_channel_group)[0]

    def decide_layer_from_means(self, work, B, S, C):
        w, b = self._wb()
        self.last_work = work
        return ops.layer_head(work.view(B, S * S, C), w, b, self.mask_channel_group)[0]

    def decide(self, x, carry=None):
        """The mask alone (what the blocks of the HIP path consume; their sparsities come from ldn_mask_to_index): forward()
        without the mean over the mask, a reduction launch per block."""
        _eval_only(self, x)
        if not self._cache_valid():
            with torch.no_grad():
                self._cache_store((self.conv.weight.detach().reshape(self.conv.weight.shape[0], -1).float().contiguous(),
                                   self.conv.bias.detach().float().contiguous()))
        w, b = self._prep
        mask, _, self.last_work = ops.spatial_masker(ops.as_nhwc(x), w, b, self.mask_channel_group, self.mask_size, False,
                                                     carry=carry, return_work=True)
        return mask


class ExpandMask(nn.Module):
    """models/utils.py:67-89.  Parameter-free; in the HIP path the dilation is part of ldn_mask_to_index
    (the dilated mask IS conv1's gather list).  forward() is kept for API parity (one mask group)."""

    def __init__(self, stride, padding=1, mask_channel_group=1):
        super().__init__()
        self.stride, self.padding, self.mask_channel_group = stride, padding, mask_channel_group

    def forward(self, x):
        if self.padding not in (0, 1) or (self.padding == 0 and self.stride != 1):
            raise LdnError("ExpandMask on the HIP path supports the (stride,1) and (1,0) dilations the blocks use")
        b, g, h, w = x.shape
        # utils.py:81: the dilation kernel is [g,g,k,k] ones -> every output group is the OR of all input groups
        u = x[:, 0] if g == 1 else x.amax(dim=1)
        if self.padding == 0:
            return (u > 0.5).unsqueeze(1).expand(b, g, h, w)
        ix = ops.mask_to_index(u.float().contiguous(), h, w, self.stride)
        return (ix.pos1 >= 0).view(b, 1, h * self.stride, w * self.stride).expand(b, g, h * self.stride, w * self.stride)


class Masker_channel_MLP(_PrepCache):
    """models/utils.py:92-131.  HIP: ldn_channel_masker (GAP + MLP + >= + active-channel list)."""

    def __init__(self, in_channels, channel_dyn_group, layers=2, reduction=16):
        super().__init__()
        assert layers in [1, 2]
        self.channel_dyn_group = channel_dyn_group
        width = max(channel_dyn_group // reduction, 16)
        self.conv = nn.Sequential(nn.Linear(in_channels, width), nn.ReLU(),
                                  nn.Linear(width, channel_dyn_group * 2, bias=True)) if layers == 2 \
            else nn.Linear(in_channels, channel_dyn_group * 2, bias=True)
        self.conv_flops = in_channels * width + width * channel_dyn_group * 2 if layers == 2 \
            else in_channels * channel_dyn_group * 2
        last = self.conv[-1] if layers == 2 else self.conv
        last.bias.data[:channel_dyn_group] = 2.0
        last.bias.data[channel_dyn_group + 1:] = -2.0
        self.layers = layers
        self._init_cache()

    def flops_for(self, x):
        return x.shape[1] * x.shape[2] * x.shape[3] + self.conv_flops

    def _weights(self):
        if not self._cache_valid():
            with torch.no_grad():
                f = lambda t: t.detach().float().contiguous()
                if self.layers == 2:
                    self._cache_store((f(self.conv[0].weight), f(self.conv[0].bias), f(self.conv[2].weight),
                                       f(self.conv[2].bias)))
                else:
                    self._cache_store((f(self.conv.weight), f(self.conv.bias), None, None))
        return self._prep

    accepts_fused_gap = True

    def lists(self, x, gran, mask_in=None, want_logits=False, gap=None):
        """-> (mask [B,G], ch_idx [B,G*gran], ch_cnt [B], logits).  `gap` = [B, splits, C] channel sums of x left by
        the producing conv's epilogue (ldn_conv_image colsum): the masker then needs no pass over x."""
        if mask_in is not None:
            return ops.channel_masker(None, None, None, None, None, self.channel_dyn_group, gran,
                                      mask_in=mask_in.float().contiguous())
        w1, b1, w2, b2 = self._weights()
        if gap is not None:
            return ops.channel_masker(None, w1, b1, w2, b2, self.channel_dyn_group, gran, want_logits=want_logits,
                                      gap_partial=gap, hw=x.shape[2] * x.shape[3])
        return ops.channel_masker(ops.as_nhwc(x), w1, b1, w2, b2, self.channel_dyn_group, gran, want_logits=want_logits)

    def forward(self, x, temperature):
        _eval_only(self, x)
        mask, _, _, _ = self.lists(x, 1)
        return mask, torch.mean(mask), self.flops_for(x)


class Masker_channel_conv_linear(_PrepCache):
    """models/utils.py:133-169: 1x1 conv + BN + ReLU on the full map, GAP, Linear.
    HIP: dense ldn_conv_image for the 1x1 conv, then ldn_channel_masker with a single linear layer."""

    def __init__(self, in_channels, channel_dyn_group, reduction=16):
        super().__init__()
        self.channel_dyn_group = channel_dyn_group
        mid = in_channels // reduction
        self.conv = nn.Sequential(conv1x1(in_channels, mid), nn.BatchNorm2d(mid), nn.ReLU())
        self.linear = nn.Linear(mid, channel_dyn_group * 2, bias=True)
        self.linear.bias.data[:channel_dyn_group] = 2.0
        self.linear.bias.data[channel_dyn_group + 1:] = -2.0
        self.masker_flops = in_channels * in_channels // reduction + mid * channel_dyn_group * 2
        self.mid = mid
        self._init_cache()

    def flops_for(self, x):
        return self.mid * x.shape[2] * x.shape[3] + self.masker_flops

    def lists(self, x, gran, mask_in=None, want_logits=False):
        if mask_in is not None:
            return ops.channel_masker(None, None, None, None, None, self.channel_dyn_group, gran,
                                      mask_in=mask_in.float().contiguous())
        if self.mid % 4 != 0:
            raise LdnError("Masker_channel_conv_linear on the HIP path needs in_channels//reduction % 4 == 0")
        if not self._cache_valid():
            with torch.no_grad():
                sc, sh = _fold_bn(self.conv[1])
                self._cache_store((self.conv[0].weight.detach().reshape(self.mid, 1, -1).float().contiguous(), sc, sh,
                                   self.linear.weight.detach().float().contiguous(),
                                   self.linear.bias.detach().float().contiguous()))
        w, sc, sh, lw, lb = self._prep
        xn = ops.as_nhwc(x)
        b, h, wd, _ = xn.shape
        y = torch.empty(b, h, wd, self.mid, device=x.device, dtype=torch.float32)
        ops.conv_image(xn, w, sc, sh, y, relu=1)
        return ops.channel_masker(y, lw, lb, None, None, self.channel_dyn_group, gran, want_logits=want_logits)

    def forward(self, x, temperature):
        _eval_only(self, x)
        mask, _, _, _ = self.lists(x, 1)
        return mask, torch.mean(mask), self.flops_for(x)


# ------------------------------------------------------------------------------------------- block
class Bottleneck(_PrepCache):
    """models/laud_resnet.py:24-165, executed sparsely on the HIP path."""
    expansion = 4
    __constants__ = ["downsample"]

    def __init__(self, inplanes, planes, stride=1, downsample=None, group_width=1, dilation=1, norm_layer=None,
                 spatial_mask_channel_group=1, channel_dyn_granularity=1, output_size=56,
                 mask_spatial_granularity=1, dyn_mode="both", channel_masker="conv_linear",
                 channel_masker_layers=2, reduction=16):
        super().__init__()
        assert dyn_mode in ["channel", "spatial", "both", "layer"]
        assert channel_masker in ["conv_linear", "MLP"]
        self.dyn_mode = dyn_mode
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        width = int(planes * (64 / 64.)) * group_width
        assert channel_dyn_granularity <= width
        channel_dyn_group = width // channel_dyn_granularity
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, group_width, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.width = width
        self.channel_dyn_granularity = channel_dyn_granularity

        self.conv1_flops_per_pixel = inplanes * width
        self.conv2_flops_per_pixel = width * width * 9 // self.conv2.groups
        self.conv3_flops_per_pixel = width * planes * self.expansion
        if self.downsample is not None:
            self.downsample_flops = inplanes * planes * self.expansion

        self.output_size = output_size
        self.mask_spatial_granularity = mask_spatial_granularity
        self.mask_size = self.output_size // self.mask_spatial_granularity if dyn_mode != "layer" else 1
        self.masker_spatial = None
        self.masker_channel = None
        if dyn_mode in ["spatial", "layer", "both"]:
            self.masker_spatial = Masker_spatial(inplanes, spatial_mask_channel_group, self.mask_size)
            self.mask_expander2 = ExpandMask(stride=1, padding=0, mask_channel_group=spatial_mask_channel_group)
            self.mask_expander1 = ExpandMask(stride=stride, padding=1, mask_channel_group=spatial_mask_channel_group)
        if dyn_mode in ["channel", "both"]:
            if channel_masker == "conv_linear":
                self.masker_channel = Masker_channel_conv_linear(inplanes, channel_dyn_group, reduction=reduction)
            else:
                self.masker_channel = Masker_channel_MLP(inplanes, channel_dyn_group, layers=channel_masker_layers,
                                                         reduction=reduction)
        # build-only hooks (default off)
        self.forced_spatial_mask = None
        self.forced_channel_mask = None
        # The residual stream may be updated IN PLACE (inactive pixels / layers then cost no HBM traffic), but only when the
        # caller owns the input tensor: ResNet.forward passes inplace=True for its own intermediates.  The module-level
        # default never mutates its input (reference semantics for Bottleneck.forward, hooks and feature taps).
        self.inplace_residual = False
        # how a channel-mode block is executed: "gather" = per-image channel-subset convs (MACs skipped), "dense" =
        # shared-weight convs over multi-image row tiles with the mask applied to the conv outputs (no MACs skipped),
        # "auto" = dense on maps of <= 64 pixels, where streaming a private weight subset per image costs more than the
        # skipped MACs save (measured) -- the per-stage decision the reference's latency predictor encodes
        self.channel_exec = "auto"
        self._init_cache()

    # ---- folded, device-resident parameters -------------------------------------------------
    def _prepare(self, device):
        if self.conv2.groups != 1 or self.conv2.dilation[0] != 1:
            raise LdnError("HIP path: grouped / dilated conv2 (ResNeXt, dilated ResNet) is not built")
        with torch.no_grad():
            W = self.width
            p = {}
            p["w1"] = self.conv1.weight.detach().reshape(W, 1, -1).float().contiguous()
            w2 = self.conv2.weight.detach().float()
            w3 = self.conv3.weight.detach().float().reshape(-1, W)
            if self.dyn_mode in ("channel", "both"):
                # k-major [taps][cin][cout]: the per-image input-channel gather becomes a ROW gather
                p["w2"] = w2.permute(2, 3, 1, 0).reshape(9, W, W).contiguous()
                p["w3"] = w3.t().reshape(1, W, -1).contiguous()
            else:
                p["w2"] = w2.permute(0, 2, 3, 1).reshape(W, 9, W).contiguous()
                p["w3"] = w3.reshape(-1, 1, W).contiguous()
            p["s1"], p["t1"] = _fold_bn(self.bn1)
            p["s2"], p["t2"] = _fold_bn(self.bn2)
            p["s3"], p["t3"] = _fold_bn(self.bn3)
            # conv3 carries bn3's scale in its weights and is called with scale=None: the kernel then starts its
            # accumulators from the residual tile (include/ldn_hip.h, "scale == NULL")
            w3s = w3 * p["s3"].view(-1, 1)
            if self.dyn_mode in ("channel", "both"):
                p["w3"] = w3s.t().reshape(1, W, -1).contiguous()
            else:
                p["w3"] = w3s.reshape(-1, 1, W).contiguous()
            if self.downsample is not None:
                dconv, dbn = self.downsample[0], self.downsample[1]
                p["wd"] = dconv.weight.detach().reshape(dconv.out_channels, 1, -1).float().contiguous()
                p["sd"], p["td"] = _fold_bn(dbn)
                p["ds_stride"] = dconv.stride[0]
            if self.dyn_mode in ("channel", "both"):
                # Channel algebra: a masked channel k of conv1's output is the CONSTANT
                # c1[k] = relu(t1[k]) (mask applied before BN, laud_resnet.py:116-118).  Writing
                # h1 = u1 + c1 with u1 = 0 on masked channels makes conv2 = W2[A,A] (*) u1 + (W2 (*) c1),
                # whose second term does not depend on the image: 16 border classes x W shifts.
                c1 = torch.relu(p["t1"])
                c2 = torch.relu(p["t2"])
                wc = torch.einsum("okyx,k->oyx", self.conv2.weight.detach().float(), c1)  # [W,3,3]
                tab = torch.empty(16, W, device=wc.device)
                for cls in range(16):
                    rb, cb = cls // 4, cls % 4
                    ys = [ky for ky in range(3) if not ((ky == 0 and rb & 1) or (ky == 2 and rb & 2))]
                    xs = [kx for kx in range(3) if not ((kx == 0 and cb & 1) or (kx == 2 and cb & 2))]
                    v = wc[:, ys][:, :, xs].sum(dim=(1, 2))
                    tab[cls] = p["t2"] + p["s2"] * v
                p["c1"], p["c2"], p["t2_tab"] = c1.contiguous(), c2.contiguous(), tab.contiguous()
                bias3 = self.conv3.weight.detach().reshape(-1, W).float() @ c2
                p["t3c"] = (p["t3"] + p["s3"] * bias3).contiguous()
            self._cache_store({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in p.items()})
        return self._prep

    # ---- execution ----------------------------------------------------------------------------
    def _dense_ix(self, B, Ho, Wo, dev):
        """Index lists of an all-active batch (every pixel of every image), cached per shape: the packed-row machinery of the
        spatial mode then is a dense convolution whose M tiles span images."""
        key = (B, Ho, Wo, self.stride, str(dev))
        cache = self.__dict__.setdefault("_dense_ix_cache", {})
        if key not in cache:
            cache[key] = ops.mask_to_index(torch.ones(B, 1, 1, device=dev), Ho, Wo, self.stride)
        return cache[key]

    def _run_channel_dense(self, x, p, gap_in=None):
        """Channel mode without gathers: conv1/conv2 run over all channels with shared (n-major) weights on row tiles that
        span images, their outputs u = relu(bn(.)) - c are zeroed on the masked channels of each image (exactly what the
        gathered form stores / skips), conv3 reads the zero-filled u2.  Same channel algebra, same results."""
        B, Cin, Hi, Wi = x.shape
        W, gran = self.width, self.channel_dyn_granularity
        Ho, Wo = (Hi - 1) // self.stride + 1, (Wi - 1) // self.stride + 1
        xn = ops.as_nhwc(x)
        if gap_in is not None and getattr(self.masker_channel, "accepts_fused_gap", False):   # GAP partials left by the producer
            mask, _, cnt, _ = self.masker_channel.lists(x, gran, mask_in=self.forced_channel_mask, gap=gap_in)
        else:
            mask, _, cnt, _ = self.masker_channel.lists(x, gran, mask_in=self.forced_channel_mask)
        # [B,1,W]: every group's decision repeated over its gran channels (one small copy; repeat_interleave is a 12 us index kernel)
        chm = (mask.unsqueeze(2).expand(-1, -1, gran).reshape(mask.shape[0], 1, -1) if gran > 1 else mask.unsqueeze(1))
        dev = x.device
        if "w2_nk" not in p:
            with torch.no_grad():
                p["w2_nk"] = self.conv2.weight.detach().float().permute(0, 2, 3, 1).reshape(W, 9, W).contiguous().to(dev)
                p["w3_nk"] = (self.conv3.weight.detach().float().reshape(-1, W) * p["s3"].view(-1, 1).to(self.conv3.weight.device)
                              ).reshape(-1, 1, W).contiguous().to(dev)
        ix = self._dense_ix(B, Ho, Wo, dev)
        x2d = xn.reshape(B * Hi * Wi, Cin)
        fused_mask = ops.dense_kernel_ok() and Cin % 32 == 0 and W % 32 == 0
        chm2d = chm.reshape(B, W).contiguous()
        h1 = torch.empty(ix.cap1, W, device=dev, dtype=torch.float32)
        if fused_mask:   # k_dense: the per-image channel mask and the post-ReLU constant are epilogue terms (no pass over h1)
            ops.conv_rows(x2d, p["w1"], p["s1"], p["t1"], h1, taps=1, m_cap=ix.cap1, relu=1, post_sub=p["c1"], chan_mask=chm2d,
                          rows_per_image=Hi * Wi)
        else:
            ops.conv_packed(x2d, p["w1"], p["s1"], p["t1"], h1, taps=1, m_cap=ix.cap1, post_sub=p["c1"], relu=1)
            h1.view(B, -1, W).mul_(chm)
        h2 = torch.empty(ix.cap3, W, device=dev, dtype=torch.float32)
        if fused_mask and 9 in ops.DENSE_TAPS and ops.DENSE_CHANNEL_3X3:
            ops.conv_rows(h1, p["w2_nk"], p["s2"], p["t2_tab"], h2, a_rows=ix.nbr, taps=9, m_cap=ix.cap3, pix_map=ix.idx3,
                          geom=(Hi, Wi, Ho, Wo, self.stride), post_sub=p["c2"], relu=1, chan_mask=chm2d, rows_per_image=Ho * Wo)
        else:
            ops.conv_packed(h1, p["w2_nk"], p["s2"], p["t2_tab"], h2, a_map=ix.nbr, taps=9, m_cap=ix.cap3, pix_map=ix.idx3,
                            geom=(Hi, Wi, Ho, Wo, self.stride), post_sub=p["c2"], relu=1)
            h2.view(B, -1, W).mul_(chm)
        cout = p["w3_nk"].shape[0]
        if self.downsample is not None:
            # (round 6: the projection on the side stream next to conv1 / conv2, as the gathered execution below does, was measured here and does
            # not pay -- stage 4's first block 644 -> 658 us: these launches are matrix-bound and fill the chip on their own)
            identity = torch.empty(B, Ho, Wo, cout, device=dev, dtype=torch.float32)
            self._shortcut(xn, p, identity)
            out = identity
        else:
            identity = xn
            out = xn if self._inplace else torch.empty_like(xn)
        ops.conv_rows(h2, p["w3_nk"], None, p["t3c"], out.view(B * Ho * Wo, cout), taps=1, m_cap=ix.cap3, relu=1,
                      residual2d=identity.view(B * Ho * Wo, cout))
        self.last_channel_mask = mask
        self.last_gap = None
        self.last_channel_cnt = cnt
        return ops.from_nhwc(out), mask

    use_fused_head = True    # conv1 on k_head (False: the general ldn_conv_image with out_format 1)
    fused_head_widths = tuple(int(t) for t in os.environ.get("LDN_HEAD_WIDTHS", "64,128,256").split(","))   # ... for these widths when a block runs on its own (with the pipelined fragment reads k_head is ahead at every width: 13.73 -> 13.62 ms)
    use_fused_tail = True    # class-level switch (A/B measurements): False keeps the three-launch gathered execution

    use_strided_tail = os.environ.get("LDN_TAIL_STRIDE2", "1") != "0"   # the fused tail on the stride-2 first blocks (A/B switch)
    use_folded_projection = os.environ.get("LDN_FOLD_PROJ", "1") != "0"   # stage 1's first block: the projection shortcut inside conv3 (A/B switch)

    def _tail_eligible(self, Hi, Wi, Ho, Wo, cout):
        """The fused conv2 -> conv3 launch (ldn_bottleneck_tail) covers: bf16x3 arithmetic, stride 1 or 2, an even channel
        granularity, widths 64 / 128 / 256, output maps at most 256 wide.  Everything else keeps the three-launch execution."""
        st = self.stride
        return (self.use_fused_tail and self.channel_exec in ("auto", "fused") and ops.fused_kernel_ok()
                and (ops.get_math_mode() != "fp32" or self.conv1.in_channels % 32 == 0)      # (the fp32 form has k_head as its only conv1)
                and (st == 1 or (st == 2 and self.use_strided_tail)) and (Ho, Wo) == ((Hi - 1) // st + 1, (Wi - 1) // st + 1)
                and self.channel_dyn_granularity % 2 == 0
                and self.width in (64, 128, 256) and Wo <= 256 and cout % 64 == 0
                and ops.bottleneck_tail_splits(Hi, Wi, self.width, st) > 0)    # LDS / slice-pipeline limits, decided by the library

    use_smallmap = os.environ.get("LDN_SMALLMAP", "1") != "0"   # the one-launch bottleneck on maps of at most 64 pixels (A/B switch)

    def _smallmap_eligible(self, Cin, Hi, Wi):
        """The whole block as ONE launch, one workgroup per image (ldn_bottleneck_smallmap): bf16x3 arithmetic, stride 1, identity
        shortcut, an even channel granularity, a map of at most 64 pixels whose activations fit the workgroup's LDS at this width
        (stage 4 of the ResNets: 7x7 x 512)."""
        cout = self.conv3.out_channels
        return (self.use_smallmap and self.channel_exec in ("auto", "smallmap") and ops.get_math_mode() == "bf16x3"
                and self.stride == 1 and self.downsample is None and Cin == cout and self.channel_dyn_granularity % 2 == 0
                and Hi * Wi <= 64 and ops.bottleneck_smallmap_fits(Hi, Wi, Cin, self.width, cout))

    def _run_channel_smallmap(self, x, p, gap_in=None, want_gap=False):
        B, Cin, Hi, Wi = x.shape
        gran = self.channel_dyn_granularity
        xn = ops.as_nhwc(x)
        if gap_in is not None and getattr(self.masker_channel, "accepts_fused_gap", False):
            mask, idx, cnt, _ = self.masker_channel.lists(x, gran, mask_in=self.forced_channel_mask, gap=gap_in)
        else:
            mask, idx, cnt, _ = self.masker_channel.lists(x, gran, mask_in=self.forced_channel_mask)
        w2p, w3p = self.tail_weights(p)
        out = xn if self._inplace else torch.empty_like(xn)
        gap_out = torch.empty(B, 2, Cin, device=x.device, dtype=torch.float32) if want_gap else None
        ops.bottleneck_smallmap(xn, p["w1s"], w2p, w3p, idx, cnt, p["s1"], p["t1"], p["c1"], p["s2"], p["t2_tab"], p["c2"], p["t3c"], out,
                                residual=xn, colsum=gap_out)
        self.last_channel_mask = mask
        self.last_gap = gap_out
        self.last_channel_cnt = cnt
        return ops.from_nhwc(out), mask

    def tail_weights(self, p):
        """conv2 / conv3 weights in the pre-split pair-interleaved layouts of ldn_bottleneck_tail (built once, cached with the
        other folded parameters).  In the fp32 math mode: their fp32 twins (ldn_bottleneck_*_f32), kept under their own keys."""
        if ops.get_math_mode() == "fp32":
            if "w2p32" not in p:
                with torch.no_grad():
                    dev = p["s3"].device
                    p["w2p32"] = ops.pack_w2_pairs(self.conv2.weight.detach().float().to(dev), f32=True)
                    w3 = self.conv3.weight.detach().float().reshape(-1, self.width).to(dev) * p["s3"].view(-1, 1)
                    p["w3p32"] = ops.pack_w3_pairs(w3, f32=True)
                    p["w1s32"] = ops.pack_w1_split(self.conv1.weight.detach().float().reshape(self.width, -1).to(dev), f32=True)
            return p["w2p32"], p["w3p32"]
        if "w2p" not in p:
            with torch.no_grad():
                dev = p["s3"].device
                p["w2p"] = ops.pack_w2_pairs(self.conv2.weight.detach().float().to(dev))
                w3 = self.conv3.weight.detach().float().reshape(-1, self.width).to(dev) * p["s3"].view(-1, 1)
                p["w3p"] = ops.pack_w3_pairs(w3)
                p["w1s"] = ops.pack_w1_split(self.conv1.weight.detach().float().reshape(self.width, -1).to(dev))
        return p["w2p"], p["w3p"]

    @staticmethod
    def _w1s_key():
        return "w1s32" if ops.get_math_mode() == "fp32" else "w1s"

    def _shortcut(self, xn, p, identity):
        """Projection shortcut (laud_resnet.py:138-141).  Maps of fewer than 96 pixels run as ONE list of strided pixel rows that
        spans the batch (M tiles then hold rows of several images) instead of per-image tiles that would be mostly empty."""
        B, Hi, Wi, Cin = xn.shape
        _, Ho, Wo, cout = identity.shape
        if Ho * Wo < 96 or (ops.dense_kernel_ok() and Cin % 32 == 0 and cout % 32 == 0):
            # one list of strided pixel rows over the batch (k_dense in bf16x3 mode: pre-split shared weights, 256-row tiles)
            ops.conv_rows(xn.reshape(B * Hi * Wi, Cin), p["wd"], p["sd"], p["td"], identity.view(B * Ho * Wo, cout), taps=1,
                          m_cap=B * Ho * Wo, a_rows=self._ds_rows(B, Hi, Wi, Ho, Wo, p["ds_stride"], xn.device), relu=0)
        else:
            ops.conv_image(xn, p["wd"], p["sd"], p["td"], identity, stride=p["ds_stride"], relu=0)

    def _run_channel(self, x, p, gap_in=None, want_gap=False):
        B, Cin, Hi, Wi = x.shape
        W, gran = self.width, self.channel_dyn_granularity
        Ho, Wo = (Hi - 1) // self.stride + 1, (Wi - 1) // self.stride + 1
        # dense execution uses the packed-row machinery, whose index set hard-codes Hi = Ho*stride on square maps: odd maps in
        # front of a stride-2 block (208 / 240 px inputs: 13x13, 15x15) and non-square maps stay on the gather path, which
        # takes the geometry explicitly
        dense_ok = Hi == Ho * self.stride and Wi == Wo * self.stride
        if self._smallmap_eligible(Cin, Hi, Wi):
            return self._run_channel_smallmap(x, p, gap_in, want_gap)
        if self.channel_exec == "dense" and not dense_ok:
            raise LdnError(f"Bottleneck: channel_exec='dense' needs Hi == Ho*stride (got {Hi}x{Wi} -> {Ho}x{Wo})")
        if dense_ok and (self.channel_exec == "dense" or (self.channel_exec == "auto" and Ho * Wo <= 64)):
            return self._run_channel_dense(x, p, gap_in)
        xn = ops.as_nhwc(x)
        if gap_in is not None and getattr(self.masker_channel, "accepts_fused_gap", False):
            mask, idx, cnt, _ = self.masker_channel.lists(x, gran, mask_in=self.forced_channel_mask, gap=gap_in)
        else:
            mask, idx, cnt, _ = self.masker_channel.lists(x, gran, mask_in=self.forced_channel_mask)
        dev = x.device
        cout = p["w3"].shape[2]
        side = None
        fold_proj = (self.downsample is not None and self.use_folded_projection and ops.get_math_mode() == "bf16x3" and self.stride == 1 and p.get("ds_stride") == 1
                     and self.use_fused_head and self.width in self.fused_head_widths and self._tail_eligible(Hi, Wi, Ho, Wo, cout)
                     and ops.bottleneck_tail_proj_fits(Hi, Wi, W, Cin))
        if fold_proj:
            # stage 1's first block: the projection shortcut is 64 more K values of conv3 (ldn_bottleneck_tail_proj) -- no projection
            # launch, no identity tensor written and read back; conv1's launch leaves its input pre-split for it
            if "wdp" not in p:
                with torch.no_grad():
                    p["wdp"] = ops.pack_w3_pairs(p["wd"].reshape(cout, Cin) * p["sd"].view(-1, 1))
                    p["t3cd"] = (p["t3c"] + p["td"]).contiguous()
            w2p, w3p = self.tail_weights(p)
            h1 = torch.empty(B, Hi, Wi, W, device=dev, dtype=torch.float32)
            xs = ops.x_split_buffer(B * Hi * Wi, Cin, dev)
            out = torch.empty(B, Ho, Wo, cout, device=dev, dtype=torch.float32)
            ops.bottleneck_head(xn, p["w1s"], idx, cnt, p["s1"], p["t1"], p["c1"], h1, x_split=xs)
            gap_out = (torch.empty(B, ops.bottleneck_tail_splits(Hi, Wi, W, 1), cout, device=dev, dtype=torch.float32) if want_gap else None)
            ops.bottleneck_tail_proj(h1, w2p, w3p, idx, cnt, p["s2"], p["t2_tab"], p["c2"], p["t3cd"], xs, p["wdp"], out, colsum=gap_out)
            self.last_channel_mask = mask
            self.last_gap = gap_out
            self.last_channel_cnt = cnt
            return ops.from_nhwc(out), mask
        if self.downsample is not None:
            # the projection shortcut only depends on x: it runs on a side stream next to conv1 / conv2 and is joined
            # before conv3 (a fork/join that hipGraph capture records as such)
            identity = torch.empty(B, Ho, Wo, cout, device=dev, dtype=torch.float32)
            if _USE_SIDE_STREAM:
                cur = torch.cuda.current_stream(dev)
                side = _side_stream(dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self._shortcut(xn, p, identity)
            else:
                self._shortcut(xn, p, identity)
            out = identity
        else:
            identity = xn
            out = xn if self._inplace else torch.empty_like(xn)
        h1 = torch.empty(B, Hi, Wi, W, device=dev, dtype=torch.float32)
        if self._tail_eligible(Hi, Wi, Ho, Wo, cout):
            # bf16x3 arithmetic, stride 1, even granularity: conv1 writes h1 pre-split, then ONE launch runs conv2 -> conv3
            # (h2 never exists in memory, every K slice of h1 is staged once for all nine taps)
            w2p, w3p = self.tail_weights(p)
            # k_head (deep staging ring, one workgroup per CU) pays where an image fills a workgroup and K is long (stage 3);
            # the early stages stream many short blocks and stay on the general kernel, two workgroups per CU (measured)
            if (self.use_fused_head and Cin % 32 == 0 and self.width in self.fused_head_widths) or ops.get_math_mode() == "fp32":
                if ops.get_math_mode() == "fp32" and Cin % 32:
                    raise LdnError("Bottleneck: the fp32 fused path needs cin % 32 == 0")
                ops.bottleneck_head(xn, p[self._w1s_key()], idx, cnt, p["s1"], p["t1"], p["c1"], h1)
            else:
                ops.conv_image(xn, p["w1"], p["s1"], p["t1"], h1, n_idx=idx, n_cnt=cnt, post_sub=p["c1"], relu=1, out_split=True)
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            gap_out = (torch.empty(B, ops.bottleneck_tail_splits(Hi, Wi, W, self.stride), cout, device=dev, dtype=torch.float32)
                       if want_gap else None)
            ops.bottleneck_tail(h1, w2p, w3p, idx, cnt, p["s2"], p["t2_tab"], p["c2"], p["t3c"], out, residual=identity,
                                colsum=gap_out, stride=self.stride)
            self.last_channel_mask = mask
            self.last_gap = gap_out
            self.last_channel_cnt = cnt
            return ops.from_nhwc(out), mask
        ops.conv_image(xn, p["w1"], p["s1"], p["t1"], h1, n_idx=idx, n_cnt=cnt, post_sub=p["c1"], relu=1)
        h2 = torch.empty(B, Ho, Wo, W, device=dev, dtype=torch.float32)
        ops.conv_image(h1, p["w2"], p["s2"], p["t2_tab"], h2, ksize=3, stride=self.stride, k_idx=idx, k_cnt=cnt,
                       kgran=gran, n_idx=idx, n_cnt=cnt, post_sub=p["c2"], relu=1)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        gap_out = torch.empty(B, (Ho * Wo + 31) // 32, cout, device=dev, dtype=torch.float32) if want_gap else None
        ops.conv_image(h2, p["w3"], None, p["t3c"], out, k_idx=idx, k_cnt=cnt, kgran=gran, relu=1,
                       residual=identity, colsum=gap_out)
        self.last_channel_mask = mask       # kept for parity tooling (bench/tests feed it to the oracle)
        self.last_gap = gap_out
        self.last_channel_cnt = cnt         # [B] active channels per image: mean(mask) = cnt.sum() / (B * width)
        return ops.from_nhwc(out), mask

    use_fused_spatial_masker = os.environ.get("LDN_FUSED_SPATIAL_MASKER", "1") != "0"   # class-level switch (A/B, tests)

    def _pool_grid(self, Hi, Wi, Ho, Wo, cout):
        """S of the S x S grid of cells whose means conv3's epilogue can leave for the next masker (None: not on the fused path): an
        identity block of ONE mask group, on the k_dense path, with in-place residual (the pixels it does not touch keep their values:
        their cells' means stay valid); spatial: the masker's own even patch grid of 4- or 16-pixel patches; layer skip (mask_size 1):
        a tiling of the square map by 4x4 or 2x2 pixel tiles (the global average pool = the mean of the tile means)."""
        ms = self.masker_spatial
        if not (self.use_fused_spatial_masker and ms.mask_channel_group == 1 and self.forced_spatial_mask is None
                and self.downsample is None and self.stride == 1 and self._inplace and cout % 128 == 0 and self.width % 8 == 0
                and ops.dense_kernel_ok()):
            return None
        S = ms.mask_size
        if S == 1:
            if Hi != Wi:
                return None
            return Hi // 4 if Hi % 4 == 0 else (Hi // 2 if Hi % 2 == 0 else None)
        if 1 < S < Hi and Hi % S == 0 and Wi % S == 0 and (Hi // S) * (Wi // S) in (4, 16) and ops.mask_plan_fits(S, S, Ho, Wo, 1):
            return S
        return None

    def _pool_grid_out(self, Ho, Wo, cout):
        """The same for a PROJECTION block (stride 1 or 2) as the producer: its output is written whole by the projection launch (every
        pixel, ReLU where the block is inactive) and rewritten at the active pixels by conv3 -- with the projection's rows listed patch by
        patch both launches can leave the mean of every cell they complete, so the block behind it decides without reading x either."""
        ms = self.masker_spatial
        if not (self.use_fused_spatial_masker and self.use_fused_projection_means and ms.mask_channel_group == 1 and self.forced_spatial_mask is None
                and self.downsample is not None and cout % 128 == 0 and self.width % 8 == 0 and ops.dense_kernel_ok()
                and ops.get_math_mode() == "bf16x3"):
            return None
        S = ms.mask_size
        if S == 1:
            if Ho != Wo:
                return None
            return Ho // 4 if Ho % 4 == 0 else (Ho // 2 if Ho % 2 == 0 else None)
        if 1 < S < Ho and Ho % S == 0 and Wo % S == 0 and (Ho // S) * (Wo // S) in (4, 16) and ops.mask_plan_fits(S, S, Ho, Wo, self.stride):
            return S
        return None

    use_fused_projection_means = os.environ.get("LDN_FUSED_PROJECTION_MEANS", "1") != "0"   # class-level switch (A/B, tests)

    def _patch_major_rows(self, B, Hi, Wi, Ho, Wo, s, S, dev):
        """(source row of x, flat output pixel) of every output pixel of the batch, listed cell by cell of the S x S grid (row-major inside a
        cell, cells row-major, images in order): the projection launch's gather / scatter lists when it leaves the cell means."""
        key = ("pm", B, Hi, Wi, Ho, Wo, s, S, str(dev))
        cache = self.__dict__.setdefault("_ds_cache", {})
        if key not in cache:
            gy, gx = Ho // S, Wo // S
            b = torch.arange(B, device=dev).view(B, 1, 1, 1, 1)
            py = torch.arange(S, device=dev).view(1, S, 1, 1, 1)
            px = torch.arange(S, device=dev).view(1, 1, S, 1, 1)
            ly = torch.arange(gy, device=dev).view(1, 1, 1, gy, 1)
            lx = torch.arange(gx, device=dev).view(1, 1, 1, 1, gx)
            oy, ox = py * gy + ly, px * gx + lx
            out_pix = ((b * Ho + oy) * Wo + ox).reshape(-1)
            src = ((b * Hi + oy * s) * Wi + ox * s).reshape(-1)
            cache[key] = (src.to(torch.int32).contiguous(), out_pix.to(torch.int32).contiguous(), out_pix.long().contiguous())
        return cache[key]

    def _consume_grid(self, Hi, Wi, Ho, Wo):
        """S of the S x S grid of cell means of its INPUT this block's masker can decide from (None: it reads x itself) -- the consumer side of
        _pool_grid, without the producer's conditions: a stride-2 / projection block at the head of a stage pools the same input cells its
        predecessor (the last identity block of the stage before) rewrote (models/utils.py:47-52: adaptive_avg_pool2d of the block INPUT)."""
        ms = self.masker_spatial
        if not (self.use_fused_spatial_masker and ms.mask_channel_group == 1 and self.forced_spatial_mask is None and ops.dense_kernel_ok()):
            return None
        S = ms.mask_size
        if S == 1:
            if Hi != Wi:
                return None
            return Hi // 4 if Hi % 4 == 0 else (Hi // 2 if Hi % 2 == 0 else None)
        if 1 < S < Hi and Hi % S == 0 and Wi % S == 0 and (Hi // S) * (Wi // S) in (4, 16, 64) and ops.mask_plan_fits(S, S, Ho, Wo, self.stride):
            return S            # (64-pixel cells: only as the 2 x 2 groups of a predecessor's 16-pixel cells, see _run_spatial)
        return None

    def _run_spatial(self, x, p):
        B, Cin, Hi, Wi = x.shape
        W = self.width
        Ho, Wo = self._out_hw(Hi, Wi)
        ms = self.masker_spatial
        G = ms.mask_channel_group
        xn = ops.as_nhwc(x)
        carry_in, self._carry_in, self.last_carry = getattr(self, "_carry_in", None), None, None
        # the fused spatial masker: an identity block whose successor decides on the same patch grid leaves the pooled
        # means of the patches it rewrites in conv3's epilogue (pool_out: its lists are then patch-major); the successor decides and
        # builds its lists from those means in ONE launch (ldn_mask_plan) -- no pass over x, no count launch
        layer = ms.mask_size == 1
        pS = self._pool_grid(Hi, Wi, Ho, Wo, p["w3"].shape[0])
        # (a projection block as the producer: the grid lives on its OUTPUT map; its pool buffer is its own, not the masker's)
        gS_ds = self._pool_grid_out(Ho, Wo, p["w3"].shape[0]) if self.downsample is not None else None
        pool_out = bool(getattr(self, "_pool_next", False)) and (pS is not None or gS_ds is not None)
        self._pool_next = False
        ix = None
        cS = pS if pS is not None else self._consume_grid(Hi, Wi, Ho, Wo)
        fresh = (cS is not None and carry_in is not None and len(carry_in) > 4 and carry_in[4] and carry_in[0] is not None
                 and tuple(carry_in[2] or ()) == (B, Hi, Wi, Cin, cS))
        # the predecessor's cells are the quarters of this masker's (the head of stage 2 of S = 4-4-2-1: 7 x 7 cells of 8 x 8 pixels behind
        # 14 x 14 of 4 x 4): the mean of a cell is the mean of its four quarters' means
        coarsen = (not fresh and not layer and pS is None and cS is not None and carry_in is not None and len(carry_in) > 4 and carry_in[4]
                   and carry_in[0] is not None and tuple(carry_in[2] or ()) == (B, Hi, Wi, Cin, 2 * cS) and Cin % 4 == 0)
        if coarsen:
            carry_in = (ops.coarsen_cell_means(carry_in[0].view(B, 2 * cS, 2 * cS, Cin), cS).view(-1),) + tuple(carry_in[1:])
            fresh = True
        # (tests / bench audits) did this block's decision come from the pooled means the previous block's conv3 epilogue left?
        self.last_fused_decision = bool(fresh) and self.forced_spatial_mask is None
        if self.forced_spatial_mask is not None:
            patch = self.forced_spatial_mask.to(device=x.device, dtype=torch.float32).contiguous()
        elif fresh and layer:
            patch = ms.decide_layer_from_means(carry_in[0], B, cS, Cin)
        elif fresh:
            patch, ix = ms.decide_from_means(carry_in[0], x.shape, (Ho, Wo), pool_out, stride=self.stride)
        elif layer and pool_out and gS_ds is None:
            patch = ms.decide_layer_pooled(x, pS)
        else:
            patch = ms.decide(x, carry=carry_in if not (carry_in is not None and len(carry_in) > 4 and layer) else None)
        dev = x.device
        # spatial_mask_channel_group > 1 (models/utils.py:27-33,74-89): group g of the OUTPUT channels has its own pixel mask.
        # ExpandMask ORs the groups (its dilation kernel is [g,g,k,k] ones), so conv1 / conv2 -- and the sparsities the
        # reference reports for them -- live on the UNION of the groups; only conv3's scatter is per group.
        union = patch[:, 0] if G == 1 else patch.amax(dim=1)
        if ix is None:
            if layer and pool_out:     # the kept images' pixels tile by tile: conv3's epilogue owns whole tiles
                gS = pS if gS_ds is None else gS_ds
                ix = ops.layer_index(union.reshape(B), Ho, Wo, self.stride, tile=(Ho // gS, Wo // gS))
            else:
                ix = ops.mask_to_index(union.contiguous(), Ho, Wo, self.stride, patch_major=pool_out)
        pool = None
        if gS_ds is not None and pool_out and ix.patch_major and self.forced_spatial_mask is None:
            cout_ = p["w3"].shape[0]
            work = torch.empty(B * gS_ds * gS_ds * cout_, device=dev, dtype=torch.float32)
            work.ldn_shape_key = (B, Ho, Wo, cout_, gS_ds)
            pool = work.view(B, gS_ds, gS_ds, cout_)
            self.last_carry = (work, ix.pre3 if layer else None, work.ldn_shape_key, None if layer else union.contiguous(), True)
        elif self.forced_spatial_mask is None and getattr(ms, "last_work", None) is not None:
            key = getattr(ms.last_work, "ldn_shape_key", None)
            if pool_out and ix.patch_major and key is not None and tuple(key) == (B, Hi, Wi, Cin, pS):
                # [4]: conv3 below refreshes the means of the cells it rewrites -> the next block decides without reading x
                pool = ms.last_work.view(B, pS, pS, Cin)
                self.last_carry = (ms.last_work, ix.pre3 if layer else None, key, None if layer else union.contiguous(), True)
            elif ms.mask_size == 1 and G == 1:
                self.last_carry = (ms.last_work, ix.pre3, key)      # layer skip: which images this block leaves unchanged, and their channel sums
            elif ms.mask_size > 1:
                self.last_carry = (ms.last_work, None, key, union.contiguous(), False)   # patch masks: the patches this block touches, every patch's pooled means
        x2d = xn.reshape(B * Hi * Wi, Cin)
        # row counts of the previous forward of THIS block (pinned memory, no synchronisation): the tile-width hint of the row kernels
        hint = getattr(self, "_rows_hint", None)
        if hint is None:
            hint = self._rows_hint = ops.RowsHint(2)
        n3, n1 = hint.get(0), hint.get(1)
        hint.update(ix.cnt)
        cout = p["w3"].shape[0]
        # round 5: h1 and h2 stay PRE-SPLIT (bf16 hi | lo per octet, the weights' layout) between the three launches -- conv1's epilogue
        # writes h1 that way, the packed 3x3 (k_rows3) and conv3 split nothing in their K loops; same values, bit-identical results
        ps = ops.rows_ps_ok(Cin, W, cout)

        def projection(out2d, groups):
            # the projection shortcut (laud_resnet.py:138-141) on the output pixels, ReLU applied where no branch output will be added
            if pool is not None and gS_ds is not None:
                # the projection's rows cell by cell: its epilogue leaves the mean of every cell (final for the inactive ones; conv3 below
                # rewrites the active cells and their means)
                src_pm, out_pm, out_pm_long = self._patch_major_rows(B, Hi, Wi, Ho, Wo, p["ds_stride"], gS_ds, dev)
                ops.conv_rows(x2d, p["wd"], p["sd"], p["td"], out2d, a_rows=src_pm, taps=1, m_cap=ix.cap3, relu=2,
                              relu_if_neg=ix.pos3.view(-1)[out_pm_long].contiguous(), out_rows=out_pm, pool=pool, pool_grid=(gS_ds, gS_ds, Ho, Wo))
            else:
                ds_rows = self._ds_rows(B, Hi, Wi, Ho, Wo, p["ds_stride"], dev)
                for ig, _, cs in groups:   # ReLU directly where the (pixel, group) is inactive: no branch output is added there
                    ops.conv_rows(x2d, p["wd"][cs], p["sd"][cs], p["td"][cs], out2d[:, cs], a_rows=ds_rows, taps=1, m_cap=ix.cap3,
                                  relu=2, relu_if_neg=ig.pos3)

        side, out2d_ds = None, None
        if self.downsample is not None and G == 1 and _USE_SIDE_STREAM:
            # round 6: it only depends on x and the lists -- on the side stream next to conv1 / the 3x3, joined in front of conv3 (as the channel path
            # does; the first block of every stage ran it between the 3x3 and conv3)
            out2d_ds = torch.empty(ix.cap3, cout, device=dev, dtype=torch.float32)
            cur = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                projection(out2d_ds, [(ix, None, slice(0, cout))])
        h1 = torch.empty(ix.cap1, W, device=dev, dtype=torch.float32)
        h2 = torch.empty(ix.cap3, W, device=dev, dtype=torch.float32)
        if ps:
            ops.conv_rows_ps(x2d, p["w1"], p["s1"], p["t1"], h1, out_presplit=True, a_rows=ix.idx1, m_count=ix.cnt[1:2], m_cap=ix.cap1, rows_hint=n1)
            ops.conv3x3_rows_ps(h1, ix.nbr, p["w2"], p["s2"], p["t2"], h2, m_count=ix.cnt[0:1], m_cap=ix.cap3, out_presplit=True, rows_hint=n3)
        else:
            ops.conv_rows(x2d, p["w1"], p["s1"], p["t1"], h1, a_rows=ix.idx1, taps=1, m_count=ix.cnt[1:2], m_cap=ix.cap1, rows_hint=n1)
            ops.conv_rows(h1, p["w2"], p["s2"], p["t2"], h2, a_rows=ix.nbr, taps=9, m_count=ix.cnt[0:1], m_cap=ix.cap3, rows_hint=n3)
        if G == 1:
            groups = [(ix, None, slice(0, cout))]
        else:
            if cout % (4 * G) != 0:
                raise LdnError("HIP path: spatial_mask_channel_group must divide the output channels into multiples of 4")
            groups = []
            ar = torch.arange(ix.cap3, device=dev, dtype=torch.int32)
            for g in range(G):
                ig = ops.mask_to_index(patch[:, g].contiguous(), Ho, Wo, self.stride)
                # packed h2 row (union list) of every pixel of this group's list; entries past the device-side count are unused
                rows = torch.where(ar < ig.cnt[0], ix.pos3[ig.idx3.clamp(0, ix.cap3 - 1).long()], torch.full_like(ar, -1))
                groups.append((ig, rows.contiguous(), slice(g * (cout // G), (g + 1) * (cout // G))))
        if self.downsample is not None:
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
                out2d = out2d_ds
            else:
                out2d = torch.empty(ix.cap3, cout, device=dev, dtype=torch.float32)
                projection(out2d, groups)
            resid = out2d
        elif self._inplace:
            resid = out2d = x2d          # x >= 0 (post-ReLU) inside the network: inactive pixels pass through
        else:
            resid, out2d = x2d, torch.relu(x2d)
        for ig, rows, cs in groups:
            if ps and (cs.stop - cs.start) % 64 == 0:
                ops.conv_rows_ps(h2, p["w3"][cs], None, p["t3"][cs], out2d[:, cs], a_presplit=True, a_rows=rows, m_count=ig.cnt[0:1],
                                 m_cap=ix.cap3, relu=1, out_rows=ig.idx3, residual2d=resid[:, cs], rows_hint=n3 if G == 1 else None,
                                 pool=pool, pool_grid=((pS if gS_ds is None else gS_ds,) * 2 + (Ho, Wo)) if pool is not None else None)
                continue
            ops.conv_rows(h2 if not ps else ops.unsplit_rows(h2), p["w3"][cs], None, p["t3"][cs], out2d[:, cs], a_rows=rows, taps=1, m_count=ig.cnt[0:1],
                          m_cap=ix.cap3, relu=1, out_rows=ig.idx3, residual2d=resid[:, cs], rows_hint=n3 if G == 1 else None, pool=pool,
                          pool_grid=((pS if gS_ds is None else gS_ds,) * 2 + (Ho, Wo)) if pool is not None else None)
        self.last_spatial_mask = patch
        if G > 1:   # sparsity of conv3 = mean over ALL group masks (Masker_spatial, utils.py:61); conv2 / conv1 = the union's
            ix.stats = torch.cat((patch.mean().reshape(1), ix.stats[1:]))
        return ops.from_nhwc(out2d.view(B, Ho, Wo, cout)), patch, ix

    def _run_both(self, x, p):
        """dyn_mode 'both' (laud_resnet.py:101-103): packed pixel lists (spatial mask) x per-image channel lists.  With several
        spatial mask groups (models/utils.py:27-33,74-89) conv1 / conv2 live on the UNION of the group masks (ExpandMask ORs the
        groups) and conv3 is scattered per output-channel group, exactly as in spatial mode."""
        B, Cin, Hi, Wi = x.shape
        W, gran = self.width, self.channel_dyn_granularity
        Ho, Wo = self._out_hw(Hi, Wi)
        ms = self.masker_spatial
        G = ms.mask_channel_group
        xn = ops.as_nhwc(x)
        dev = x.device
        cmask, idx, cnt, _ = self.masker_channel.lists(x, gran, mask_in=self.forced_channel_mask)
        if self.forced_spatial_mask is not None:
            patch = self.forced_spatial_mask.to(device=dev, dtype=torch.float32).contiguous()
        else:
            patch = ms.decide(x)
        union = patch[:, 0] if G == 1 else patch.amax(dim=1)
        ix = ops.mask_to_index(union.contiguous(), Ho, Wo, self.stride)
        x2d = xn.reshape(B * Hi * Wi, Cin)
        geom = (Hi, Wi, Ho, Wo, self.stride)
        h1 = torch.empty(ix.cap1, W, device=dev, dtype=torch.float32)
        ops.conv_packed(x2d, p["w1"], p["s1"], p["t1"], h1, B=B, row_prefix=ix.pre1, m_cap=Hi * Wi, a_map=ix.idx1, taps=1,
                        n_idx=idx, n_cnt=cnt, post_sub=p["c1"], relu=1)
        h2 = torch.empty(ix.cap3, W, device=dev, dtype=torch.float32)
        ops.conv_packed(h1, p["w2"], p["s2"], p["t2_tab"], h2, B=B, row_prefix=ix.pre3, m_cap=Ho * Wo, a_map=ix.nbr, taps=9,
                        pix_map=ix.idx3, geom=geom, k_idx=idx, k_cnt=cnt, kgran=gran, n_idx=idx, n_cnt=cnt,
                        post_sub=p["c2"], relu=1)
        cout = p["w3"].shape[2]
        if G == 1:
            # (a_map: the packed h2 row of every list entry -- the identity for one group)
            groups = [(ix, None, slice(0, cout), p["w3"], p["t3c"])]
        else:
            if cout % (4 * G) != 0:
                raise LdnError("HIP path: spatial_mask_channel_group must divide the output channels into multiples of 4")
            if "w3_groups" not in p or len(p["w3_groups"]) != G:
                p["w3_groups"] = [p["w3"][:, :, g * (cout // G):(g + 1) * (cout // G)].contiguous() for g in range(G)]
            groups = []
            ar = torch.arange(ix.cap3, device=dev, dtype=torch.int32)
            for g in range(G):
                ig = ops.mask_to_index(patch[:, g].contiguous(), Ho, Wo, self.stride)
                rows = torch.where(ar < ig.cnt[0], ix.pos3[ig.idx3.clamp(0, ix.cap3 - 1).long()], torch.full_like(ar, -1))
                cs = slice(g * (cout // G), (g + 1) * (cout // G))
                groups.append((ig, rows.contiguous(), cs, p["w3_groups"][g], p["t3c"][cs]))
        if self.downsample is not None:
            out2d = torch.empty(ix.cap3, cout, device=dev, dtype=torch.float32)
            ds_rows = self._ds_rows(B, Hi, Wi, Ho, Wo, p["ds_stride"], dev)
            for ig, _, cs, _, _ in groups:   # ReLU directly where the (pixel, group) is inactive: no branch output is added there
                ops.conv_rows(x2d, p["wd"][cs], p["sd"][cs], p["td"][cs], out2d[:, cs], a_rows=ds_rows, taps=1, m_cap=ix.cap3,
                              relu=2, relu_if_neg=ig.pos3)
            resid = out2d
        elif self._inplace:
            resid = out2d = x2d
        else:
            resid, out2d = x2d, torch.relu(x2d)
        for ig, rows, cs, w3g, t3g in groups:
            ops.conv_packed(h2, w3g, None, t3g, out2d[:, cs], B=B, row_prefix=ig.pre3, m_cap=Ho * Wo, a_map=rows, taps=1,
                            out_map=ig.idx3, k_idx=idx, k_cnt=cnt, kgran=gran, relu=1, residual2d=resid[:, cs])
        self.last_channel_mask, self.last_spatial_mask = cmask, patch
        if G > 1:   # sparsity of conv3 = mean over ALL group masks (Masker_spatial, utils.py:61); conv2 / conv1 = the union's
            ix.stats = torch.cat((patch.mean().reshape(1), ix.stats[1:]))
        return ops.from_nhwc(out2d.view(B, Ho, Wo, cout)), cmask, ix

    def _out_hw(self, Hi, Wi):
        """Output map of the block for an Hi x Wi input.  The masks are interpolated to the ACTUAL output map (what the
        reference's detection backbone does, lad_mmdet_resnet.py:274; on the classifier's own input size it is output_size, as
        laud_resnet.py:106 hard-codes): non-square and non-224 inputs work.  The packed-row index build needs Hi = Ho * stride."""
        s = self.stride
        if Hi % s or Wi % s:
            raise LdnError(f"Bottleneck: a {Hi}x{Wi} input is not a multiple of the block's stride {s} (spatial / layer / both modes)")
        return Hi // s, Wi // s

    def _ds_rows(self, B, Hi, Wi, Ho, Wo, s, dev):
        key = (B, Hi, Wi, s, str(dev))
        cache = self.__dict__.setdefault("_ds_cache", {})
        if key not in cache:
            b = torch.arange(B, device=dev).view(B, 1, 1)
            y = torch.arange(Ho, device=dev).view(1, Ho, 1) * s
            xx = torch.arange(Wo, device=dev).view(1, 1, Wo) * s
            cache[key] = ((b * Hi + y) * Wi + xx).reshape(-1).to(torch.int32).contiguous()
        return cache[key]

    def run_dynamic(self, x, gap_in=None, want_gap=False, defer_stats=False, inplace=None):
        """Execute the block on the HIP path.  gap_in / want_gap: fused global-average-pool hand-off between
        consecutive channel-mode blocks (the conv3 epilogue leaves the channel sums the next masker needs).  Returns (out, stats[4] = {s3, s2, s1, channel sparsity} as a device
        tensor).  The FLOPs bookkeeping is separate (flops_terms) so a whole network can do it once, vectorised."""
        _eval_only(self, x)
        p = self._prep if self._cache_valid() else self._prepare(x.device)
        # in-place residual update: only on request (ResNet.forward, for tensors it owns) or by explicit opt-in on the module
        self._inplace = bool(self.inplace_residual if inplace is None else inplace) and self.downsample is None
        if self.dyn_mode == "both":
            out, cmask, ix = self._run_both(x, p)
            stats = torch.cat((ix.stats, cmask.mean().reshape(1)))
        elif self.dyn_mode == "channel":
            out, cmask = self._run_channel(x, p, gap_in, want_gap)
            if defer_stats:   # a whole network turns the per-image channel counts of all its blocks into sparsities at once
                return out, (self.last_channel_cnt, float(x.shape[0] * self.width))
            stats = torch.ones(4, device=x.device)
            stats[3] = cmask.mean()
        else:
            out, patch, ix = self._run_spatial(x, p)
            if defer_stats:   # (the channel sparsity 1 is appended to all such blocks at once: a fill and a cat per block otherwise)
                return out, ix.stats
            stats = torch.cat((ix.stats, torch.ones(1, device=x.device)))
        return out, stats

    def flops_terms(self, x_shape):
        """Shape-only constants of the bookkeeping of laud_resnet.py:112-147:
        (masker flops, conv1, conv2, conv3, downsample) with conv1 counted at the INPUT resolution."""
        _, _, hi, wi = x_shape
        px_in = hi * wi
        px_out = ((hi - 1) // self.stride + 1) * ((wi - 1) // self.stride + 1)
        probe = torch.empty(x_shape, device="meta")
        masker = 0
        if self.masker_channel is not None:
            masker += self.masker_channel.flops_for(probe)
        if self.masker_spatial is not None:
            masker += self.masker_spatial.flops_for(probe)
        ds = self.downsample_flops * px_out if self.downsample is not None else 0
        return (masker, self.conv1_flops_per_pixel * px_in, self.conv2_flops_per_pixel * px_out,
                self.conv3_flops_per_pixel * px_out, ds)

    def forward(self, x, temperature=1.0):
        x, s3_list, s2_list, s1_list, cs_list, perc_list, flops = x
        out, stats = self.run_dynamic(x)
        s3, s2, s1, cs = stats[0], stats[1], stats[2], stats[3]
        masker, c1, c2, c3, ds = self.flops_terms(x.shape)

        # FLOPs bookkeeping, laud_resnet.py:112-147
        dense = masker + c1 + c2 + c3 + ds
        sparse = masker + c1 * cs * s1
        sparse = sparse + c2 * cs ** 2 * s2
        sparse = sparse + c3 * cs * s3
        sparse = sparse + ds
        flops = flops + sparse
        perc = sparse / dense

        def push(lst, v):
            v = v.reshape(1)
            return v if lst is None else torch.cat((lst, v), dim=0)

        return (out, push(s3_list, s3), push(s2_list, s2), push(s1_list, s1), push(cs_list, cs),
                push(perc_list, perc), flops)


# ------------------------------------------------------------------------------------------- model
class ResNet(nn.Module):
    """models/laud_resnet.py:167-363."""

    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, width_mult=1., input_size=224,
                 spatial_mask_channel_group=[1, 1, 1, 1], mask_spatial_granularity=[1, 1, 1, 1],
                 channel_dyn_granularity=[1, 1, 1, 1], dyn_mode=["both", "both", "both", "both"],
                 channel_masker=["MLP", "MLP", "MLP", "MLP"], channel_masker_layers=[1, 1, 1, 1],
                 reduction_ratio=[16, 16, 16, 16], lr_mult=1.0, **kwargs):
        super().__init__()
        self.dyn_mode = dyn_mode
        assert lr_mult is not None
        self.lr_mult = lr_mult
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = int(64 * width_mult)
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(
                replace_stride_with_dilation))
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        dil = [False] + list(replace_stride_with_dilation)
        for i, (mult, stride, down) in enumerate(((64, 1, 4), (128, 2, 8), (256, 2, 16), (512, 2, 32))):
            setattr(self, f"layer{i + 1}", self._make_layer(
                block, int(mult * width_mult), layers[i], stride=stride, dilate=dil[i],
                output_size=input_size // down, spatial_mask_channel_group=spatial_mask_channel_group[i],
                mask_spatial_granularity=mask_spatial_granularity[i],
                channel_dyn_granularity=channel_dyn_granularity[i], dyn_mode=dyn_mode[i],
                channel_masker=channel_masker[i], channel_masker_layers=channel_masker_layers[i],
                reduction_ratio=reduction_ratio[i]))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(int(512 * width_mult * block.expansion), num_classes)

        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d) and "masker" not in name:
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
        # The network owns its intermediate activations: forward() asks blocks without a downsample branch to update the
        # residual stream in place (their input is post-ReLU, hence >= 0).  Set False to keep every block input intact
        # (forward hooks / feature taps on block inputs).
        self.inplace_residual = True
        self._tap = None

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False, output_size=56,
                    spatial_mask_channel_group=1, mask_spatial_granularity=1, channel_dyn_granularity=1,
                    dyn_mode="both", channel_masker="MLP", channel_masker_layers=1, reduction_ratio=16):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        common = dict(group_width=self.groups, norm_layer=norm_layer, output_size=output_size,
                      spatial_mask_channel_group=spatial_mask_channel_group,
                      mask_spatial_granularity=mask_spatial_granularity,
                      channel_dyn_granularity=channel_dyn_granularity, dyn_mode=dyn_mode,
                      channel_masker=channel_masker, channel_masker_layers=channel_masker_layers,
                      reduction=reduction_ratio)
        mods = [block(inplanes=self.inplanes, planes=planes, stride=stride, downsample=downsample,
                      dilation=previous_dilation, **common)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            mods.append(block(self.inplanes, planes, dilation=self.dilation, **common))
        return nn.ModuleList(mods)

    def forward(self, x, temperature):
        _eval_only(self, x)
        in_shape = tuple(x.shape)
        x = self._stem_forward(x)
        x, stats, sizes = self._run_blocks(x, gap0=self.__dict__.pop("_stem_gap", None))
        st, perc, flops = self._forward_stats(stats, in_shape, x.device)   # st [n_blocks, 4] = s3, s2, s1, cs
        s3, s2, s1, cs = st[:, 0], st[:, 1], st[:, 2], st[:, 3]

        x = self.avgpool(x)
        x = torch.flatten(x, 1)
        x = self.fc(x)
        split = lambda v: list(torch.split(v, sizes))
        return x, split(s3), split(s2), split(s1), split(cs), perc, flops

    def _stem_forward(self, x):
        # static stem (laud_resnet.py:318-324): plain library ops, channels-last so the blocks see NHWC rows
        x = x.contiguous(memory_format=torch.channels_last)
        # eval-mode stem = conv with bn1's scale folded into its weights, then max-pool, then bn1's shift and the ReLU on the POOLED
        # map: relu(maxpool(y + t)) == relu(maxpool(y) + t) exactly (adding a per-channel constant and ReLU are monotone, the pool's
        # padding is -inf) -- three full-resolution passes fewer than conv, bn, relu (a conv bias is a separate full-size pass in MIOpen)
        w, b = self._folded_stem()
        if self._stem_fused_ok(x):
            # one launch: conv 7x7 -> max-pool -> + shift -> ReLU; the full-resolution conv output never exists (ldn_stem_conv_pool)
            first = self.layer1[0]
            if (self.use_stem_gap and first.dyn_mode == "channel" and first.forced_channel_mask is None and self._tap is None
                    and getattr(first.masker_channel, "accepts_fused_gap", False)):
                # ... and leaves the channel sums of its output for the first block's channel masker (no pass over x for its GAP)
                y, self._stem_gap = ops.stem_conv_pool(ops.as_nhwc(x), self._stem_frag, b, self.conv1.out_channels, want_gap=True)
                return ops.from_nhwc(y)
            x = ops.from_nhwc(ops.stem_conv_pool(ops.as_nhwc(x), self._stem_frag, b, self.conv1.out_channels))
        else:
            x = F.conv2d(x, w, None, self.conv1.stride, self.conv1.padding)
            x = self.maxpool(x).add_(b.view(1, -1, 1, 1)).relu_()

        return x

    use_stem_gap = os.environ.get("LDN_STEM_GAP", "1") != "0"   # the fused stem leaves the first channel masker's GAP partials (A/B switch)

    def _run_blocks(self, x, stage_outs=None, gap0=None):
        """The four stages on the HIP path.  Returns (x, per-block stats, blocks per stage); stage_outs (a list) receives every
        stage's output map (the detection backbone's feature taps)."""
        # dynamic blocks: each returns its 4 sparsities as a device vector; the FLOPs bookkeeping of
        # laud_resnet.py:112-147,329-347 is done ONCE below on [n_blocks] vectors (no per-block scalar kernels)
        stats = []
        blocks = [blk for i in range(4) for blk in getattr(self, f"layer{i + 1}")]
        sizes = [len(getattr(self, f"layer{i + 1}")) for i in range(4)]
        ends = set(itertools.accumulate(sizes))        # a stage's output = the output of its last block
        step_id = self._step_id = getattr(self, "_step_id", 0) + 1
        gap = gap0          # channel sums of the stem's output, when the fused stem left them (first block's masker)
        j = -1
        while j + 1 < len(blocks):
            j += 1
            blk = blocks[j]
            nxt = blocks[j + 1] if j + 1 < len(blocks) else None
            # a run of stride-1 channel-mode blocks on a map that fits one workgroup (stage 3): ONE launch, image by image
            n_run = self._chain_len(blocks, j, x, gap)
            if n_run >= 2:
                x, run_stats, gap = self._run_chain(blocks[j:j + n_run], x, gap)
                stats.extend(run_stats)
                j += n_run - 1
                nxt = blocks[j + 1] if j + 1 < len(blocks) else None
                if not (nxt is not None and nxt.dyn_mode == "channel" and nxt.forced_channel_mask is None
                        and getattr(nxt.masker_channel, "accepts_fused_gap", False)):
                    gap = None
                if stage_outs is not None and j + 1 in ends:
                    stage_outs.append(x)
                continue
            # a channel-mode block leaves the GAP partials of its output for the next block's MLP masker
            want_gap = (nxt is not None and blk.dyn_mode == "channel" and nxt.dyn_mode == "channel"
                        and getattr(nxt.masker_channel, "accepts_fused_gap", False) and nxt.forced_channel_mask is None)
            if self._tap is not None:     # debug tap (bench / tests): sees every block's input; off by default
                self._tap(j, blk, x)
            # layer skip: the images the previous block skipped reach this block unchanged -> their channel sums (the masker's global
            # average pool) are carried over instead of re-read (one full read of x per block otherwise)
            prev = blocks[j - 1] if j > 0 else None
            # (a carry of FRESH cell means -- conv3's epilogue left the mean of every cell of the block's output -- also serves the stride-2 /
            # projection block at the head of the next stage: its masker pools its INPUT, which is that output)
            fresh_means = prev is not None and len(getattr(prev, "last_carry", None) or ()) > 4 and bool(prev.last_carry[4])
            blk._carry_in = (prev.last_carry if (self.use_layer_carry and prev is not None and blk.dyn_mode in ("layer", "spatial") and prev.dyn_mode == blk.dyn_mode
                                                 and ((blk.stride == 1 and blk.downsample is None) or (fresh_means and self.use_stage_carry)) and blk.forced_spatial_mask is None
                                                 # the producer must leave the images it skips UNCHANGED: a block with a projection
                                                 # shortcut / stride 2 turns them into relu(downsample(x))
                                                 and ((prev.stride == 1 and prev.downsample is None) or fresh_means)
                                                 and getattr(prev, "_carry_step", -1) == step_id) else None)
            blk._pool_next = (self.use_layer_carry and nxt is not None and blk.dyn_mode in ("spatial", "layer") and nxt.dyn_mode == blk.dyn_mode
                              and ((nxt.stride == 1 and nxt.downsample is None) or self.use_stage_carry) and nxt.forced_spatial_mask is None
                              and (nxt.masker_spatial.mask_size == blk.masker_spatial.mask_size
                                   or (self.use_stage_carry and blk.masker_spatial.mask_size == 2 * nxt.masker_spatial.mask_size))
                              and nxt.masker_spatial.mask_channel_group == 1)
            x, st = blk.run_dynamic(x, gap_in=gap, want_gap=want_gap, defer_stats=True, inplace=self.inplace_residual)
            blk._carry_step = step_id if getattr(blk, "last_carry", None) is not None else -1
            gap = getattr(blk, "last_gap", None) if want_gap else None
            stats.append(st)
            if stage_outs is not None and j + 1 in ends:
                stage_outs.append(x)
        return x, stats, sizes

    use_stage_carry = os.environ.get("LDN_STAGE_CARRY", "1") != "0"   # the fused masker across a stage boundary (class-level switch: A/B, tests)

    # ---- chained execution of a run of blocks (ldn_bottleneck_chain)
    use_chain = os.environ.get("LDN_CHAIN", "1") != "0"     # class-level switch (A/B measurements, tests): False launches every block on its own

    chain_max_blocks = int(os.environ.get("LDN_CHAIN_MAX", "64"))   # longest run per launch (tuning)
    use_fused_stem = os.environ.get("LDN_FUSED_STEM", "1") != "0"    # the one-launch stem (ldn_stem_conv_pool); 0 = library conv + pool
    use_layer_carry = os.environ.get("LDN_LAYER_CARRY", "1") != "0"  # layer skip: carry the skipped images' channel sums to the next masker

    def _chain_len(self, blocks, j, x, gap):
        """Number of consecutive blocks from j that ldn_bottleneck_chain can execute as one launch (0 = none).  The run needs the
        GAP partials of its input (left by the producing block), identical shapes, MLP maskers deciding on their own, and
        everything ldn_bottleneck_tail needs; the result is bit-identical to the block-by-block execution."""
        if not self.use_chain or gap is None or self._tap is not None or not x.is_cuda:
            return 0
        B, C, H, W = x.shape
        if H * W > 256 or H * W <= 64 or gap.dim() != 3 or gap.shape[0] != B or gap.shape[2] != C:
            return 0
        first, n = blocks[j], 0
        m0 = getattr(first, "masker_channel", None)
        if not isinstance(m0, Masker_channel_MLP) or first.width not in (64, 128, 256):
            return 0
        # the library decides whether the run's phases fit one workgroup's LDS (a 16x16 map of a 256-wide layer does not: the
        # blocks then run one by one, which does fit)
        if not ops.bottleneck_chain_fits(H, W, C, first.width, m0.conv[0].out_features if m0.layers == 2 else 0, m0.channel_dyn_group):
            return 0
        for blk in blocks[j:]:
            m = blk.masker_channel
            ok = (isinstance(blk, Bottleneck) and blk.dyn_mode == "channel" and isinstance(m, Masker_channel_MLP)
                  and blk.forced_channel_mask is None and blk.downsample is None and blk.stride == 1
                  and blk.use_fused_head and blk._tail_eligible(H, W, H, W, C) and C % 32 == 0
                  and blk.conv1.in_channels == C and blk.conv3.out_channels == C
                  and blk.width == first.width and blk.channel_dyn_granularity == first.channel_dyn_granularity
                  and m.layers == first.masker_channel.layers and m.channel_dyn_group == first.masker_channel.channel_dyn_group
                  and (m.layers == 1 or m.conv[0].out_features == first.masker_channel.conv[0].out_features))
            if not ok or n >= self.chain_max_blocks:
                break
            n += 1
        return n

    def _run_chain(self, run, x, gap):
        dev = x.device
        preps = [blk._prep if blk._cache_valid() else blk._prepare(dev) for blk in run]
        mws = [blk.masker_channel._weights() for blk in run]
        f32 = ops.get_math_mode() == "fp32"
        key = (str(dev), f32) + tuple(blk._prep_gen for blk in run) + tuple(blk.masker_channel._prep_gen for blk in run)
        cache = getattr(self, "_chain_cache", None)
        if cache is None:
            cache = self._chain_cache = {}
        ent = cache.get(id(run[0]))
        if ent is None or ent[0] != key:
            rows = []
            for blk, p, mw in zip(run, preps, mws):
                w2p, w3p = blk.tail_weights(p)
                rows.append((p[blk._w1s_key()], p["s1"], p["t1"], p["c1"], w2p, w3p, p["s2"], p["t2_tab"], p["c2"], p["t3c"],
                             mw[0], mw[1], mw[2], mw[3]))
            ent = cache[id(run[0])] = (key, ops.chain_table(rows, dev), rows)     # rows: keeps the tensors alive
        first = run[0]
        m = first.masker_channel
        hidden = m.conv[0].out_features if m.layers == 2 else 0
        xn = ops.as_nhwc(x)
        work = xn if self.inplace_residual else torch.empty_like(xn)
        masks, idx, cnt, colsum = ops.bottleneck_chain(xn, work, ent[1], first.width, hidden, m.channel_dyn_group,
                                                       first.channel_dyn_granularity, gap, f32=f32)
        stats = []
        denom = float(x.shape[0] * first.width)
        for i, blk in enumerate(run):
            blk.last_channel_mask = masks[i]
            blk.last_channel_cnt = cnt[i]
            blk.last_gap = None
            stats.append((cnt[i], denom))
        run[-1].last_gap = colsum
        return ops.from_nhwc(work), stats, colsum

    # ---- FLOPs bookkeeping (laud_resnet.py:112-147, 321-356) as a function of the input SHAPE and the sparsities only
    def flops_table(self, x_shape):
        """Shape-only constants: (terms [n_blocks][5] = masker, conv1, conv2, conv3, downsample per block; static FLOPs of the
        stem, max-pool, average pool and classifier).  No tensor is touched: usable on any host (laudnet_amd.distributed
        recomputes the global-batch flops_perc / flops from all-reduced sparsities with it)."""
        _, c_in, h, w = x_shape
        k, s, pd = self.conv1.kernel_size[0], self.conv1.stride[0], self.conv1.padding[0]
        h, w = (h + 2 * pd - k) // s + 1, (w + 2 * pd - k) // s + 1
        c = self.conv1.out_channels
        static = c_in * c * h * w * k * k
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        static += c * h * w * 9
        terms = []
        for i in range(4):
            for blk in getattr(self, f"layer{i + 1}"):
                terms.append(blk.flops_terms((1, c, h, w)))
                c = blk.conv3.out_channels
                h, w = (h - 1) // blk.stride + 1, (w - 1) // blk.stride + 1
        static += self._head_flops(c)
        return terms, static

    def _head_flops(self, c):
        """adaptive average pool to 1x1 (x.shape[1] * 1 * 1 after pooling, laud_resnet.py:350) + classifier (:356)"""
        return c + c * self.fc.out_features

    def flops_from_sparsities(self, x_shape, s3, s2, s1, cs):
        """(flops_perc [n_blocks], flops) from per-block sparsities (flat [n_blocks] tensors or per-stage lists).  This is the
        ONLY place the model forms them: forward() calls it, and so does the multi-GPU gather with global-batch sparsities."""
        flat = lambda v: torch.cat([t.reshape(-1) for t in v]) if isinstance(v, (list, tuple)) else v
        s3, s2, s1, cs = (flat(v).double() for v in (s3, s2, s1, cs))   # fp64 inside: the result does not depend on summation order
        key = (str(s3.device), tuple(x_shape[1:]))
        if getattr(self, "_terms_key", None) != key:
            terms, static = self.flops_table(x_shape)
            self._terms_key = key
            self._terms = torch.tensor(terms, dtype=torch.float64, device=s3.device)   # [n_blocks, 5]
            self._static_flops = float(static)
        tm = self._terms
        sparse = tm[:, 0] + tm[:, 1] * cs * s1
        sparse = sparse + tm[:, 2] * cs ** 2 * s2
        sparse = sparse + tm[:, 3] * cs * s3
        sparse = sparse + tm[:, 4]
        perc = sparse / tm.sum(dim=1)
        flops = sparse.sum() + self._static_flops
        return perc.float(), flops.float()

    def _stem_fused_ok(self, x):
        """The one-launch stem (k_stem) covers the standard geometry in bf16x3 arithmetic; anything else (fp32 math mode, the
        reduced widths of the tiny test models, a modified stem) runs conv -> max-pool as library ops."""
        c, mp = self.conv1, self.maxpool
        ok = (self.use_fused_stem and ops.get_math_mode() == "bf16x3" and c.in_channels == 3 and c.out_channels in (32, 64)
              and c.kernel_size == (7, 7) and c.stride == (2, 2) and c.padding == (3, 3) and c.dilation == (1, 1) and c.groups == 1
              and _pair(mp.kernel_size) == (3, 3) and _pair(mp.stride) == (2, 2) and _pair(mp.padding) == (1, 1)
              and _pair(mp.dilation) == (1, 1) and not mp.ceil_mode)
        if ok and getattr(self, "_stem_frag_key", None) != self._stem_key:
            self._stem_frag = ops.pack_stem_weights(self._stem[0])
            self._stem_frag_key = self._stem_key
        return ok

    def _folded_stem(self):
        """conv1 weights scaled by bn1's eval affine (laud_resnet.py:318-320), cached until a parameter or buffer changes."""
        bn = self.bn1
        src = (self.conv1.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in src)
        if getattr(self, "_stem_key", None) != key:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = (self.conv1.weight * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
                self._stem = (w, (bn.bias - bn.running_mean * scale).contiguous())
            self._stem_key = key
        return self._stem

    use_fused_stats = os.environ.get("LDN_FUSED_STATS", "1") != "0"   # the bookkeeping of a forward as one launch (ldn_forward_stats)

    def _forward_stats(self, stats, in_shape, dev):
        """Per-block stats -> (st [n_blocks, 4], flops_perc [n_blocks], flops).  All blocks channel mode (they hand over their per-image
        channel counts) or all blocks spatial / layer (three sparsities each): ONE stack + ONE launch (ldn_forward_stats; eager PyTorch
        spent ~28 tiny kernels = 0.13 ms of a 13 ms step here).  Mixed models take the tensor-op path below."""
        chan = [isinstance(s_, tuple) for s_ in stats]
        if self.use_fused_stats and dev.type == "cuda" and (all(chan) or not any(chan)):
            key = (str(dev), tuple(in_shape[1:]))
            if getattr(self, "_terms_key", None) != key:
                terms, static = self.flops_table(in_shape)
                self._terms_key = key
                self._terms = torch.tensor(terms, dtype=torch.float64, device=dev)
                self._static_flops = float(static)
            if all(chan):
                same = len({tuple(s_[0].shape) for s_ in stats}) == 1
                if same:
                    dkey = (str(dev), tuple(s_[1] for s_ in stats))
                    if getattr(self, "_denoms_key", None) != dkey:
                        self._denoms_key = dkey
                        self._denoms = torch.tensor(dkey[1], dtype=torch.float32, device=dev)
                    cnt = torch.stack([s_[0] for s_ in stats])
                    return ops.forward_stats(self._terms, self._static_flops, cnt=cnt, denom=self._denoms)
            elif all((not isinstance(s_, tuple)) and s_.numel() == stats[0].numel() for s_ in stats):
                return ops.forward_stats(self._terms, self._static_flops, st_in=torch.stack(stats))
        st = self._stack_stats(stats, dev)
        perc, flops = self.flops_from_sparsities(in_shape, st[:, 0], st[:, 1], st[:, 2], st[:, 3])
        return st, perc, flops

    def _stack_stats(self, stats, dev):
        """Per-block stats -> [n_blocks, 4].  Channel-mode blocks hand over (per-image channel counts, B * width):
        their sparsity row is (1, 1, 1, cnt.sum() / (B * width)), formed for all such blocks with three small kernels
        instead of a mean + fill + copy per block."""
        short = [j for j, st in enumerate(stats) if not isinstance(st, tuple) and st.numel() == 3]
        if short:   # spatial / layer blocks hand over (s3, s2, s1); their channel sparsity is 1
            if len(short) == len(stats):
                st3 = torch.stack(stats)
                return torch.cat((st3, torch.ones(st3.shape[0], 1, device=dev)), dim=1)
            one = torch.ones(1, device=dev)
            stats = [torch.cat((st, one)) if (not isinstance(st, tuple) and st.numel() == 3) else st for st in stats]
        deferred = [j for j, st in enumerate(stats) if isinstance(st, tuple)]
        if not deferred:
            return torch.stack(stats)
        cs = torch.stack([stats[j][0] for j in deferred]).sum(dim=1).to(torch.float32)
        dkey = (str(dev), tuple(stats[j][1] for j in deferred))
        if getattr(self, "_denoms_key", None) != dkey:
            self._denoms_key = dkey
            self._denoms = torch.tensor(dkey[1], dtype=torch.float32, device=dev)
        cs = cs / self._denoms
        rows = torch.ones(len(deferred), 4, device=dev)
        rows[:, 3] = cs
        if len(deferred) == len(stats):
            return rows
        out, k = [], 0
        for j, st in enumerate(stats):
            if isinstance(st, tuple):
                out.append(rows[k])
                k += 1
            else:
                out.append(st)
        return torch.stack(out)

    def get_optim_policies(self):
        backbone_params, masker_params = [], []
        for name, m in self.named_modules():
            dst = masker_params if "masker" in name else backbone_params
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                dst.extend(list(m.parameters(recurse=False)))
            elif isinstance(m, nn.BatchNorm2d) or (isinstance(m, nn.BatchNorm1d) and dst is masker_params):
                dst.extend(list(m.parameters(recurse=False)))
        return [{"params": backbone_params, "lr_mult": self.lr_mult, "decay_mult": 1.0, "name": "backbone_params"},
                {"params": masker_params, "lr_mult": 1.0, "decay_mult": 1.0, "name": "masker_params"}]


def _resnet(arch, block, layers, pretrained, progress, **kwargs):
    if pretrained:
        raise LdnError("pretrained=True needs network access; load a state_dict explicitly")
    return ResNet(block, layers, **kwargs)


def uni_resnet50(pretrained=False, progress=True, **kwargs):
    return _resnet("resnet50", Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


def uni_resnet101(pretrained=False, progress=True, **kwargs):
    return _resnet("resnet101", Bottleneck, [3, 4, 23, 3], pretrained, progress, **kwargs)


class GraphedForward:
    """Replay a model's eval forward as one hipGraph (no host launch gaps between the ~250 kernels of a forward).
    The channel/spatial paths keep all data-dependent sizes on the device, so the launch sequence is static.
    `model(x, temperature)` is captured once on a side stream for a fixed input shape; calling the object copies
    the new batch into the static input buffer and replays."""

    def __init__(self, model, example_x, temperature=1.0, warmup=2):
        self.model = model
        self.static_x = example_x.clone(memory_format=torch.channels_last)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):          # library autotuning, lazy weight folding, LDS attribute set-up
                model(self.static_x, temperature)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(self.static_x, temperature)

    def __call__(self, x):
        self.static_x.copy_(x)
        self.graph.replay()
        return self.static_out
