"""LAD-RegNet on the MI355X HIP path: host-side mirror of imagenet_classification/models/laud_regnet.py.

Same class names, constructor arguments, sub-module layout (hence state_dict keys: `stem.{0,1}.*`,
`trunk_output.block{S}.block{S}-{i}.{proj.{0,1},f.{a,b,c}.{0,1},f.se.fc{1,2},f.masker_*}.*`, `fc.*`) and
`forward(x, temperature)` 7-tuple as the reference, so its checkpoints load unchanged.

Execution.  **Layer skip** (the configuration BASELINE.json names for RegNet: `dyn_mode='spatial'` with
`mask_spatial_granularity = output size`, one keep/skip bit per image and block; the reference rejects `dyn_mode='layer'` for
RegNet, laud_regnet.py:100): kept images run a (1x1) -> b (grouped 3x3) -> SE -> c (1x1) on packed rows in libldn_hip.so,
skipped images cost nothing.  **Channel mode**: the mask multiplies the outputs of a and b after their activation, so only the
active channel subsets are computed (a with an output list, b on the active run of every group, SE gathered, c with an input
list).  **General spatial / both**: the reference masks only conv c's output and the SE squeeze pools the dense conv-b output
(laud_regnet.py:194,200), so a / b / SE run on every pixel (on the active channels in 'both') and c on the packed
active pixels.  spatial_mask_channel_group > 1 raises LdnError.  No CPU / PyTorch fallback.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from ._lib import LdnError
from .laud_resnet import (Masker_channel_conv_linear, Masker_channel_MLP, Masker_spatial, ExpandMask, _PrepCache,
                          _eval_only, _fold_bn)

__all__ = ["LAD_RegNet", "BlockParams", "lad_regnet_y_400mf", "lad_regnet_y_800mf", "lad_regnet_y_1_6gf",
           "lad_regnet_y_3_2gf", "lad_regnet_y_8gf", "lad_regnet_y_16gf"]


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class ConvNormActivation(nn.Sequential):
    """conv(bias=False) -> norm -> [activation]: the container layout of torchvision 0.14's class of the same name."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, groups=1, norm_layer=nn.BatchNorm2d,
                 activation_layer=nn.ReLU):
        layers = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, (kernel_size - 1) // 2, groups=groups,
                            bias=norm_layer is None)]
        if norm_layer is not None:
            layers.append(norm_layer(out_channels))
        if activation_layer is not None:
            layers.append(activation_layer(inplace=True))
        super().__init__(*layers)
        self.out_channels = out_channels


class SqueezeExcitation(nn.Module):
    def __init__(self, input_channels, squeeze_channels):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(input_channels, squeeze_channels, 1)
        self.fc2 = nn.Conv2d(squeeze_channels, input_channels, 1)


class SimpleStemIN(ConvNormActivation):
    def __init__(self, width_in, width_out, norm_layer, activation_layer):
        super().__init__(width_in, width_out, kernel_size=3, stride=2, norm_layer=norm_layer,
                         activation_layer=activation_layer)


class BottleneckTransform(_PrepCache):
    """laud_regnet.py:74-217."""

    def __init__(self, width_in, width_out, stride, norm_layer, activation_layer, group_width, bottleneck_multiplier,
                 se_ratio, spatial_mask_channel_group=1, channel_dyn_granularity=1, output_size=56,
                 mask_spatial_granularity=1, dyn_mode="both", channel_masker="conv_linear", channel_masker_layers=2,
                 reduction=16):
        super().__init__()
        assert dyn_mode in ["channel", "spatial", "both"]
        assert channel_masker in ["conv_linear", "MLP"]
        self.dyn_mode = dyn_mode
        w_b = int(round(width_out * bottleneck_multiplier))
        g = w_b // group_width
        self.a = ConvNormActivation(width_in, w_b, kernel_size=1, stride=1, norm_layer=norm_layer,
                                    activation_layer=activation_layer)
        self.b = ConvNormActivation(w_b, w_b, kernel_size=3, stride=stride, groups=g, norm_layer=norm_layer,
                                    activation_layer=activation_layer)
        if se_ratio:
            width_se_out = int(round(se_ratio * width_in))
            self.se = SqueezeExcitation(w_b, width_se_out)
        self.c = ConvNormActivation(w_b, width_out, kernel_size=1, stride=1, norm_layer=norm_layer, activation_layer=None)
        assert channel_dyn_granularity <= w_b
        channel_dyn_group = w_b // channel_dyn_granularity
        self.conv1_flops_per_pixel = width_in * w_b
        self.conv2_flops_per_pixel = w_b * w_b * 9 // g
        self.conv3_flops_per_pixel = w_b * width_out
        self.se_flops_per_pixel = w_b * width_se_out * 2 if se_ratio else 0
        self.has_se = bool(se_ratio)
        self.stride, self.w_b, self.group_width = stride, w_b, w_b // g
        self.output_size = output_size
        self.mask_spatial_granularity = mask_spatial_granularity
        self.mask_size = self.output_size // self.mask_spatial_granularity
        self.masker_spatial = None
        self.masker_channel = None
        if dyn_mode in ["spatial", "both"]:
            self.masker_spatial = Masker_spatial(width_in, spatial_mask_channel_group, self.mask_size)
            self.mask_expander2 = ExpandMask(stride=1, padding=0, mask_channel_group=spatial_mask_channel_group)
            self.mask_expander1 = ExpandMask(stride=stride, padding=1, mask_channel_group=spatial_mask_channel_group)
        if dyn_mode in ["channel", "both"]:
            if channel_masker == "conv_linear":
                self.masker_channel = Masker_channel_conv_linear(width_in, channel_dyn_group, reduction=reduction)
            else:
                self.masker_channel = Masker_channel_MLP(width_in, channel_dyn_group, layers=channel_masker_layers,
                                                         reduction=reduction)
        self.forced_spatial_mask = None
        self.forced_channel_mask = None
        self._init_cache()

    def prepared(self, device):
        if self._cache_valid():
            return self._prep
        if not self.has_se:
            raise LdnError("HIP path: RegNet-X (no SE) is not built -- the reference itself fails on it (self.se undefined)")
        with torch.no_grad():
            w_b, gw = self.w_b, self.group_width
            p = {}
            p["wa"] = self.a[0].weight.detach().reshape(w_b, 1, -1).float().contiguous()
            p["sa"], p["ta"] = _fold_bn(self.a[1])
            p["wb"] = self.b[0].weight.detach().permute(0, 2, 3, 1).reshape(w_b, 9, gw).float().contiguous()
            if gw == 16 and w_b % 16 == 0:      # matrix-core form of conv b (bf16x3 mode): weights pre-split in MFMA fragment order
                p["wb_frag"] = ops.pack_grouped16_weights(p["wb"])
            p["sb"], p["tb"] = _fold_bn(self.b[1])
            p["wc"] = self.c[0].weight.detach().reshape(-1, 1, w_b).float().contiguous()
            p["wc_k"] = self.c[0].weight.detach().reshape(-1, w_b).float().t().reshape(1, w_b, -1).contiguous()   # k-major (channel mode)
            p["sc"], p["tc"] = _fold_bn(self.c[1])
            s = self.se.fc1.out_channels
            p["se_w1"] = self.se.fc1.weight.detach().reshape(s, w_b).float().contiguous()
            p["se_b1"] = self.se.fc1.bias.detach().float().contiguous()
            p["se_w2"] = self.se.fc2.weight.detach().reshape(w_b, s).float().contiguous()
            p["se_b2"] = self.se.fc2.bias.detach().float().contiguous()
            self._cache_store({k: v.to(device) for k, v in p.items()})
        return self._prep


def _conv_b_rows(p, f, h_a, nbr, h_b, m_count, m_cap, images=None):
    """conv b (grouped 3x3 + BN + ReLU) over packed rows: on the matrix cores for group width 16 in bf16x3 mode, else the fp32 VALU
    kernel.  images = (B, Hi, Wi, Ho, Wo, stride) when the rows are whole images in order (layer skip, dense index): maps that fit
    the LDS are staged there once per (image, group chunk) instead of being read once per tap through the neighbour table."""
    if "wb_frag" in p and ops.get_math_mode() == "bf16x3" and USE_GROUPED_MFMA:
        if (images is not None and m_count is not None and USE_GROUPED_IMAGES
                and ops.grouped16_images_fit(images[1], images[2], h_b.shape[1]) > 0):
            ops.grouped16_conv3x3_images(h_a, p["wb_frag"], p["sb"], p["tb"], h_b, m_count=m_count, images=images, relu=1)
        else:
            ops.grouped16_conv3x3_rows(h_a, nbr, p["wb_frag"], p["sb"], p["tb"], h_b, m_count=m_count, m_cap=m_cap, relu=1)
    else:
        ops.grouped_conv3x3_rows(h_a, nbr, p["wb"], f.group_width, p["sb"], p["tb"], h_b, m_count=m_count, m_cap=m_cap, relu=1)


USE_GROUPED_MFMA = os.environ.get("LDN_GROUPED_MFMA", "1") != "0"    # tuning switch (A/B)
USE_GROUPED_IMAGES = os.environ.get("LDN_GROUPED_IMAGES", "1") != "0"   # ... the whole-image form of it
USE_SE_FUSED = os.environ.get("LDN_SE_FUSED", "1") != "0"               # ... the SE block folded into conv b's epilogue and conv c's input rows


class ResBottleneckBlock(_PrepCache):
    """laud_regnet.py:221-295, executed for layer skip on the HIP path."""

    def __init__(self, width_in, width_out, stride, norm_layer, activation_layer, group_width=1,
                 bottleneck_multiplier=1.0, se_ratio=None, spatial_mask_channel_group=1, channel_dyn_granularity=1,
                 output_size=56, mask_spatial_granularity=1, dyn_mode="both", channel_masker="conv_linear",
                 channel_masker_layers=2, reduction=16):
        super().__init__()
        self.proj = None
        if (width_in != width_out) or (stride != 1):
            self.proj = ConvNormActivation(width_in, width_out, kernel_size=1, stride=stride, norm_layer=norm_layer,
                                           activation_layer=None)
        self.f = BottleneckTransform(width_in, width_out, stride, norm_layer, activation_layer, group_width,
                                     bottleneck_multiplier, se_ratio, spatial_mask_channel_group, channel_dyn_granularity,
                                     output_size, mask_spatial_granularity, dyn_mode, channel_masker, channel_masker_layers,
                                     reduction)
        self.activation = activation_layer(inplace=True)
        self.dyn_mode = dyn_mode
        self.stride = stride
        if self.proj is not None:
            self.downsample_flops = width_in * width_out
        self.inplace_residual = False
        self._init_cache()

    def _proj(self, device):
        if not self._cache_valid():
            with torch.no_grad():
                sp, tp = _fold_bn(self.proj[1])
                w = self.proj[0].weight.detach().reshape(self.proj[0].out_channels, 1, -1).float().contiguous()
                self._cache_store((w.to(device), sp.to(device), tp.to(device)))
        return self._prep

    def _ds_rows(self, B, Hi, Wi, Ho, Wo, s, dev):
        key = (B, Hi, Wi, s, str(dev))
        cache = self.__dict__.setdefault("_ds_cache", {})
        if key not in cache:
            b = torch.arange(B, device=dev).view(B, 1, 1)
            y = torch.arange(Ho, device=dev).view(1, Ho, 1) * s
            xx = torch.arange(Wo, device=dev).view(1, 1, Wo) * s
            cache[key] = ((b * Hi + y) * Wi + xx).reshape(-1).to(torch.int32).contiguous()
        return cache[key]

    def flops_terms(self, x_shape):
        """(masker, conv a, conv b, conv c, proj, se) of laud_regnet.py:179-203,286-288 (a counted at the input resolution)."""
        f = self.f
        _, _, hi, wi = x_shape
        px_in = hi * wi
        px_out = ((hi - 1) // self.stride + 1) * ((wi - 1) // self.stride + 1)
        probe = torch.empty(x_shape, device="meta")
        masker = 0
        if f.masker_channel is not None:
            masker += f.masker_channel.flops_for(probe)
        if f.masker_spatial is not None:
            masker += f.masker_spatial.flops_for(probe)
        proj = self.downsample_flops * px_out if self.proj is not None else 0
        return (masker, f.conv1_flops_per_pixel * px_in, f.conv2_flops_per_pixel * px_out,
                f.conv3_flops_per_pixel * px_out, proj, f.se_flops_per_pixel)

    def run_dynamic(self, x, inplace=None, defer_stats=False):
        """-> (out, stats[4] = s3, s2, s1, channel sparsity).  inplace: update the residual stream in place (only the owner
        of x may ask for it: LAD_RegNet.forward does for its own intermediates; the module default never mutates its input)."""
        _eval_only(self, x)
        inplace = bool(self.inplace_residual if inplace is None else inplace) and self.proj is None
        f = self.f
        if f.dyn_mode == "channel":
            return self._run_channel(x, inplace)
        if f.dyn_mode == "both" or f.mask_size != 1 or f.masker_spatial.mask_channel_group != 1:
            # (several mask groups: every group of output channels has its own mask -- also with one patch per image, where a
            # dropped image may keep the other half of its channels -- so a / b / SE run densely and c is scattered per group)
            return self._run_spatial_general(x, inplace, defer_stats)
        p = f.prepared(x.device)
        B, Cin, Hi, Wi = x.shape
        Ho = Wo = f.output_size
        if Hi != Ho * self.stride or Wi != Wo * self.stride:
            raise LdnError(f"ResBottleneckBlock: input {Hi}x{Wi} does not match output_size {Ho} * stride {self.stride}")
        dev = x.device
        xn = ops.as_nhwc(x)
        carry_in, f._carry_in, f.last_carry = getattr(f, "_carry_in", None), None, None
        if f.forced_spatial_mask is not None:
            patch = f.forced_spatial_mask.to(device=dev, dtype=torch.float32).contiguous()
        else:
            patch = f.masker_spatial.decide(x, carry=carry_in)
        ix = ops.mask_to_index(patch[:, 0].contiguous(), Ho, Wo, self.stride)
        if f.forced_spatial_mask is None and getattr(f.masker_spatial, "last_work", None) is not None:
            f.last_carry = (f.masker_spatial.last_work, ix.pre3, getattr(f.masker_spatial.last_work, "ldn_shape_key", None))   # which images this block leaves unchanged, and their channel sums
        x2d = xn.reshape(B * Hi * Wi, Cin)
        w_b = f.w_b
        hint = getattr(f, "_rows_hint", None)     # row counts of this block's previous forward: tile-width hint of the row kernels
        if hint is None:
            hint = f._rows_hint = ops.RowsHint(2)
        n3, n1 = hint.get(0), hint.get(1)
        hint.update(ix.cnt)
        h_a = torch.empty(ix.cap1, w_b, device=dev, dtype=torch.float32)
        ops.conv_rows(x2d, p["wa"], p["sa"], p["ta"], h_a, a_rows=ix.idx1, taps=1, m_count=ix.cnt[1:2], m_cap=ix.cap1, rows_hint=n1)
        h_b = torch.empty(ix.cap3, w_b, device=dev, dtype=torch.float32)
        # the SE block folded into its neighbours (whole kept images in order): conv b leaves the channel sums of its output, the SE head
        # turns them into a gate per kept image, conv c multiplies its input rows by it in flight -- three launches instead of six
        gate = None
        if (USE_SE_FUSED and "wb_frag" in p and ops.get_math_mode() == "bf16x3" and USE_GROUPED_MFMA and USE_GROUPED_IMAGES
                and ops.conv_rows_gated_fits(w_b, Ho * Wo) and ops.grouped16_images_fit(Hi, Wi, w_b) > 0):
            gap = ops.grouped16_conv3x3_images_gap(h_a, p["wb_frag"], p["sb"], p["tb"], h_b, m_count=ix.cnt[0:1],
                                                   images=(B, Hi, Wi, Ho, Wo, self.stride), relu=1)
            gate = ops.se_gate_slots(gap, ix.cnt[0:1], Ho * Wo, p["se_w1"], p["se_b1"], p["se_w2"], p["se_b2"])
        else:
            _conv_b_rows(p, f, h_a, ix.nbr, h_b, ix.cnt[0:1], ix.cap3, images=(B, Hi, Wi, Ho, Wo, self.stride))
            ops.se_packed(h_b, ix.pre3, p["se_w1"], p["se_b1"], p["se_w2"], p["se_b2"], Ho * Wo)
        cout = p["wc"].shape[0]
        if self.proj is not None:
            # (round 6: on a side stream next to conv a / b, as LAUD-ResNet's spatial path runs its projections, this launch costs more than it
            # hides here -- 3.22 -> 3.31 ms per forward: RegNet's launches are 30-70 us, the fork / join and the contention outweigh the overlap)
            wp, sp, tp = self._proj(dev)
            out2d = torch.empty(ix.cap3, cout, device=dev, dtype=torch.float32)
            ops.conv_rows(x2d, wp, sp, tp, out2d, a_rows=self._ds_rows(B, Hi, Wi, Ho, Wo, self.stride, dev), taps=1,
                          m_cap=ix.cap3, relu=2, relu_if_neg=ix.pos3)
            resid = out2d
        elif inplace:
            resid = out2d = x2d
        else:
            resid, out2d = x2d, torch.relu(x2d)
        if gate is not None:
            ops.conv_rows_gated(h_b, p["wc"], p["sc"], p["tc"], out2d, gate, Ho * Wo, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1,
                                out_rows=ix.idx3, residual2d=resid)
        else:
            ops.conv_rows(h_b, p["wc"], p["sc"], p["tc"], out2d, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=1,
                          out_rows=ix.idx3, residual2d=resid, rows_hint=n3)
        f.last_spatial_mask = patch
        # (defer_stats: the caller appends the channel sparsity 1 to all blocks at once -- a fill and a cat per block otherwise)
        stats = ix.stats if defer_stats else torch.cat((ix.stats, torch.ones(1, device=dev)))
        return ops.from_nhwc(out2d.view(B, Ho, Wo, cout)), stats

    def _run_spatial_general(self, x, inplace, defer_stats=False):
        """dyn_mode 'spatial' with patch masks, and 'both' (laud_regnet.py:164-217).  The reference masks ONLY conv c's output
        spatially (:200): a and b run on every pixel and the SE squeeze pools the dense b output (:194), so what the pixel mask can
        skip exactly is conv c (+ the residual update) on the inactive pixels -- a / b / SE are executed densely (in 'both' mode on
        the image's active channels, as in channel mode), c on the packed rows of the active pixels, scattered back.  (Layer skip,
        one patch per image, is the special case where a dropped image needs no a / b / SE at all: run_dynamic above.)"""
        f = self.f
        p = f.prepared(x.device)
        B, Cin, Hi, Wi = x.shape
        s = self.stride
        if Hi % s or Wi % s:
            raise LdnError(f"ResBottleneckBlock: a {Hi}x{Wi} input is not a multiple of the block's stride {s}")
        Ho, Wo = Hi // s, Wi // s
        dev = x.device
        xn = ops.as_nhwc(x)
        w_b = f.w_b
        both = f.dyn_mode == "both"
        if f.forced_spatial_mask is not None:
            patch = f.forced_spatial_mask.to(device=dev, dtype=torch.float32).contiguous()
        else:
            patch = f.masker_spatial.decide(x)
        G = f.masker_spatial.mask_channel_group
        cout = p["wc"].shape[0]
        # spatial_mask_channel_group > 1 (laud_regnet.py:145-147,172-177,198 with models/utils.py:27-33,74-89): group g of the OUTPUT
        # channels of conv c has its own pixel mask; ExpandMask ORs the groups, so the sparsities the reference reports for conv b / a
        # are those of the UNION of the group masks, the one of conv c is the mean over all group masks.
        union = patch[:, 0] if G == 1 else patch.amax(dim=1)
        ix = ops.mask_to_index(union.contiguous(), Ho, Wo, s)                # union list and the three sparsities
        if G == 1:
            groups = [(ix, slice(0, cout))]
        else:
            if cout % (4 * G) != 0:
                raise LdnError("HIP path: spatial_mask_channel_group must divide the output channels into multiples of 4")
            groups = [(ops.mask_to_index(patch[:, g].contiguous(), Ho, Wo, s), slice(g * (cout // G), (g + 1) * (cout // G))) for g in range(G)]
            ix.stats = torch.cat((patch.mean().reshape(1), ix.stats[1:]))
        x2d = xn.reshape(B * Hi * Wi, Cin)
        if both:
            if w_b % 8 != 0:
                raise LdnError("HIP path: LAD-RegNet 'both' mode needs a bottleneck width that is a multiple of 8")
            gran = w_b // f.masker_channel.channel_dyn_group
            cmask, idx, cnt, _ = f.masker_channel.lists(x, gran, mask_in=f.forced_channel_mask)
            h_a = torch.empty(B, Hi, Wi, w_b, device=dev, dtype=torch.float32)
            ops.conv_image(xn, p["wa"], p["sa"], p["ta"], h_a, n_idx=idx, n_cnt=cnt, relu=1)
            h_b = torch.empty(B, Ho, Wo, w_b, device=dev, dtype=torch.float32)
            ops.grouped_conv3x3_image(h_a, p["wb"], f.group_width, idx, cnt, p["sb"], p["tb"], h_b, stride=s, relu=1)
            h_b2d = h_b.view(B * Ho * Wo, w_b)
            ops.se_packed(h_b2d, self._img_prefix(B, Ho * Wo, dev), p["se_w1"], p["se_b1"], p["se_w2"], p["se_b2"], Ho * Wo,
                          ch_idx=idx, ch_cnt=cnt)
        else:
            dense = self._dense_index(B, Ho, Wo, dev)                          # every pixel: the neighbour table of conv b
            h_a = torch.empty(B * Hi * Wi, w_b, device=dev, dtype=torch.float32)
            ops.conv_rows(x2d, p["wa"], p["sa"], p["ta"], h_a, taps=1, m_cap=B * Hi * Wi)
            h_b2d = torch.empty(B * Ho * Wo, w_b, device=dev, dtype=torch.float32)
            _conv_b_rows(p, f, h_a, dense.nbr, h_b2d, dense.cnt[0:1], B * Ho * Wo, images=(B, Hi, Wi, Ho, Wo, s))
            ops.se_packed(h_b2d, self._img_prefix(B, Ho * Wo, dev), p["se_w1"], p["se_b1"], p["se_w2"], p["se_b2"], Ho * Wo)
        if self.proj is not None:
            wp, sp, tp = self._proj(dev)
            out2d = torch.empty(ix.cap3, cout, device=dev, dtype=torch.float32)
            ds_rows = self._ds_rows(B, Hi, Wi, Ho, Wo, s, dev)
            for ig, cs_ in groups:   # ReLU directly where the (pixel, group) is inactive: no branch output is added there
                ops.conv_rows(x2d, wp[cs_], sp[cs_], tp[cs_], out2d[:, cs_], a_rows=ds_rows, taps=1, m_cap=ix.cap3, relu=2,
                              relu_if_neg=ig.pos3)
            resid = out2d
        elif inplace:
            resid = out2d = x2d
        else:
            resid, out2d = x2d, torch.relu(x2d)
        if both:   # c: gathered input channels (per image) x packed active pixels (per output-channel group)
            for g, (ig, cs_) in enumerate(groups):
                if G == 1:
                    wck = p["wc_k"]
                else:   # sliced once per module (setdefault would evaluate the slice + copy on every forward)
                    gk = f"wc_k_grp{g}of{G}"
                    if gk not in p:
                        p[gk] = p["wc_k"][:, :, cs_].contiguous()
                    wck = p[gk]
                ops.conv_packed(h_b2d, wck, p["sc"][cs_], p["tc"][cs_], out2d[:, cs_], B=B, row_prefix=ig.pre3, m_cap=Ho * Wo, a_map=ig.idx3,
                                taps=1, out_map=ig.idx3, k_idx=idx, k_cnt=cnt, kgran=gran, relu=1, residual2d=resid[:, cs_])
            f.last_channel_mask = cmask
            cs = cmask.mean().reshape(1)
        else:
            for ig, cs_ in groups:
                ops.conv_rows(h_b2d, p["wc"][cs_], p["sc"][cs_], p["tc"][cs_], out2d[:, cs_], a_rows=ig.idx3, taps=1, m_count=ig.cnt[0:1],
                              m_cap=ix.cap3, relu=1, out_rows=ig.idx3, residual2d=resid[:, cs_])
            cs = None if defer_stats else torch.ones(1, device=dev)
        f.last_spatial_mask = patch
        return ops.from_nhwc(out2d.view(B, Ho, Wo, cout)), (ix.stats if cs is None else torch.cat((ix.stats, cs)))

    def _dense_index(self, B, Ho, Wo, dev):
        key = (B, Ho, Wo, self.stride, str(dev))
        cache = self.__dict__.setdefault("_dense_ix_cache", {})
        if key not in cache:
            cache[key] = ops.mask_to_index(torch.ones(B, 1, 1, device=dev), Ho, Wo, self.stride)
        return cache[key]

    def _run_channel(self, x, inplace):
        """dyn_mode 'channel' (laud_regnet.py:160-170,182-189): the mask multiplies the outputs of a and b AFTER conv+BN+ReLU, so
        masked channels are exact zeros and computing only the active subsets is exact: a = 1x1 with an output-channel list,
        b = grouped 3x3 over the active channels of each group, SE on the kept channels, c = 1x1 with an input-channel list
        (+ residual + ReLU fused)."""
        f = self.f
        p = f.prepared(x.device)
        B, Cin, Hi, Wi = x.shape
        s = self.stride
        Ho, Wo = (Hi - 1) // s + 1, (Wi - 1) // s + 1
        dev = x.device
        xn = ops.as_nhwc(x)
        w_b = f.w_b
        gran = w_b // f.masker_channel.channel_dyn_group
        if w_b % 8 != 0:
            raise LdnError("HIP path: LAD-RegNet channel mode needs a bottleneck width that is a multiple of 8")
        mask, idx, cnt, _ = f.masker_channel.lists(x, gran, mask_in=f.forced_channel_mask)
        h_a = torch.empty(B, Hi, Wi, w_b, device=dev, dtype=torch.float32)
        ops.conv_image(xn, p["wa"], p["sa"], p["ta"], h_a, n_idx=idx, n_cnt=cnt, relu=1)
        h_b = torch.empty(B, Ho, Wo, w_b, device=dev, dtype=torch.float32)
        ops.grouped_conv3x3_image(h_a, p["wb"], f.group_width, idx, cnt, p["sb"], p["tb"], h_b, stride=s, relu=1)
        ops.se_packed(h_b.view(B * Ho * Wo, w_b), self._img_prefix(B, Ho * Wo, dev), p["se_w1"], p["se_b1"], p["se_w2"],
                      p["se_b2"], Ho * Wo, ch_idx=idx, ch_cnt=cnt)
        cout = p["wc"].shape[0]
        if self.proj is not None:
            wp, sp, tp = self._proj(dev)
            identity = torch.empty(B, Ho, Wo, cout, device=dev, dtype=torch.float32)
            ops.conv_image(xn, wp, sp, tp, identity, stride=s, relu=0)
            out = identity
        else:
            identity = xn
            out = xn if inplace else torch.empty_like(xn)
        ops.conv_image(h_b, p["wc_k"], p["sc"], p["tc"], out, k_idx=idx, k_cnt=cnt, kgran=gran, relu=1, residual=identity)
        f.last_channel_mask = mask
        stats = torch.ones(4, device=dev)
        stats[3] = mask.mean()
        return ops.from_nhwc(out), stats

    def _img_prefix(self, B, rows, dev):
        key = (B, rows, str(dev))
        cache = self.__dict__.setdefault("_prefix_cache", {})
        if key not in cache:
            cache[key] = (torch.arange(B + 1, device=dev, dtype=torch.int32) * rows).contiguous()
        return cache[key]

    def forward(self, x, temperature):
        x, s3_list, s2_list, s1_list, cs_list, perc_list, flops = x
        out, st = self.run_dynamic(x)
        s3, s2, s1, cs = st[0], st[1], st[2], st[3]
        masker, c1, c2, c3, proj, se = self.flops_terms(x.shape)
        dense = masker + c1 + c2 + c3 + proj
        sparse = masker + c1 * cs * s1
        sparse = sparse + c2 * cs ** 2 * s2
        sparse = sparse + c3 * cs * s3
        sparse = sparse + proj
        flops = flops + se + sparse
        perc = sparse / dense

        def push(lst, v):
            v = v.reshape(1)
            return v if lst is None else torch.cat((lst, v), dim=0)

        return (out, push(s3_list, s3), push(s2_list, s2), push(s1_list, s1), push(cs_list, cs), push(perc_list, perc), flops)


class AnyStage(nn.Sequential):
    """laud_regnet.py:298-354."""

    def __init__(self, width_in, width_out, stride, depth, block_constructor, norm_layer, activation_layer, group_width,
                 bottleneck_multiplier, se_ratio=None, stage_index=0, **dyn):
        super().__init__()
        for i in range(depth):
            block = block_constructor(width_in if i == 0 else width_out, width_out, stride if i == 0 else 1, norm_layer,
                                      activation_layer, group_width, bottleneck_multiplier, se_ratio, **dyn)
            self.add_module(f"block{stage_index}-{i}", block)

    def forward(self, x, temperature):
        for layer in self.children():
            x = layer(x, temperature)
        return x


class BlockParams:
    """laud_regnet.py:357-465."""

    def __init__(self, depths, widths, group_widths, bottleneck_multipliers, strides, se_ratio=None):
        self.depths, self.widths, self.group_widths = depths, widths, group_widths
        self.bottleneck_multipliers, self.strides, self.se_ratio = bottleneck_multipliers, strides, se_ratio

    @classmethod
    def from_init_params(cls, depth, w_0, w_a, w_m, group_width, bottleneck_multiplier=1.0, se_ratio=None, **kwargs):
        QUANT, STRIDE = 8, 2
        if w_a < 0 or w_0 <= 0 or w_m <= 1 or w_0 % 8 != 0:
            raise ValueError("Invalid RegNet settings")
        widths_cont = torch.arange(depth) * w_a + w_0
        block_capacity = torch.round(torch.log(widths_cont / w_0) / math.log(w_m))
        block_widths = (torch.round(torch.divide(w_0 * torch.pow(w_m, block_capacity), QUANT)) * QUANT).int().tolist()
        num_stages = len(set(block_widths))
        splits = [w != wp or r != rp for w, wp, r, rp in zip(block_widths + [0], [0] + block_widths, block_widths + [0],
                                                             [0] + block_widths)]
        stage_widths = [w for w, t in zip(block_widths, splits[:-1]) if t]
        stage_depths = torch.diff(torch.tensor([d for d, t in enumerate(splits) if t])).int().tolist()
        strides = [STRIDE] * num_stages
        mults = [bottleneck_multiplier] * num_stages
        gws = [group_width] * num_stages
        widths = [int(w * b) for w, b in zip(stage_widths, mults)]
        gws = [min(g, w_bot) for g, w_bot in zip(gws, widths)]
        ws_bot = [_make_divisible(w_bot, g) for w_bot, g in zip(widths, gws)]
        stage_widths = [int(w_bot / b) for w_bot, b in zip(ws_bot, mults)]
        return cls(depths=stage_depths, widths=stage_widths, group_widths=gws, bottleneck_multipliers=mults,
                   strides=strides, se_ratio=se_ratio)

    def _get_expanded_params(self):
        return zip(self.widths, self.strides, self.depths, self.group_widths, self.bottleneck_multipliers)


class LAD_RegNet(nn.Module):
    """laud_regnet.py:468-672."""
    use_layer_carry = __import__("os").environ.get("LDN_LAYER_CARRY", "1") != "0"

    def __init__(self, block_params, num_classes=1000, stem_width=32, stem_type=None, block_type=None, norm_layer=None,
                 activation=None, input_size=224, spatial_mask_channel_group=[1, 1, 1, 1],
                 mask_spatial_granularity=[1, 1, 1, 1], channel_dyn_granularity=[1, 1, 1, 1],
                 dyn_mode=["both", "both", "both", "both"], channel_masker=["MLP", "MLP", "MLP", "MLP"],
                 channel_masker_layers=[1, 1, 1, 1], reduction_ratio=[16, 16, 16, 16], lr_mult=1.0, **kwargs):
        super().__init__()
        stem_type = stem_type or SimpleStemIN
        norm_layer = norm_layer or nn.BatchNorm2d
        block_type = block_type or ResBottleneckBlock
        activation = activation or nn.ReLU
        self.dyn_mode = dyn_mode
        assert lr_mult is not None
        self.lr_mult = lr_mult
        self.stem = stem_type(3, stem_width, norm_layer, activation)
        current_width = stem_width
        blocks = []
        for i, (width_out, stride, depth, group_width, bottleneck_multiplier) in enumerate(
                block_params._get_expanded_params()):
            blocks.append((f"block{i + 1}", AnyStage(
                current_width, width_out, stride, depth, block_type, norm_layer, activation, group_width,
                bottleneck_multiplier, block_params.se_ratio, stage_index=i + 1,
                spatial_mask_channel_group=spatial_mask_channel_group[i],
                channel_dyn_granularity=channel_dyn_granularity[i], output_size=input_size // (2 ** (i + 2)),
                mask_spatial_granularity=mask_spatial_granularity[i], dyn_mode=dyn_mode[i],
                channel_masker=channel_masker[i], channel_masker_layers=channel_masker_layers[i],
                reduction=reduction_ratio[i])))
            current_width = width_out
        self.trunk_output = nn.Sequential(OrderedDict(blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(in_features=current_width, out_features=num_classes)
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d) and "masker" not in name:
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, mean=0.0, std=math.sqrt(2.0 / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear) and "masker" not in name:
                nn.init.normal_(m.weight, mean=0.0, std=0.01)
                nn.init.zeros_(m.bias)
        self._tap = None
        self.inplace_residual = True   # the network owns its intermediate activations (see laud_resnet.ResNet)

    def blocks(self):
        return [blk for stage in self.trunk_output.children() for blk in stage.children()]

    use_fused_stem = os.environ.get("LDN_FUSED_STEM", "1") != "0"    # the one-launch stem (ldn_stem3_conv); 0 = library conv, BN, ReLU

    def _stem_forward(self, x):
        """Static stem (laud_regnet.py:59-71, :380): conv 3x3 stride 2 -> BN -> ReLU.  The standard geometry in bf16x3 arithmetic is
        ONE launch (ldn_stem3_conv: BN scale folded into the weights, shift + ReLU in the epilogue); anything else (fp32 math
        mode, the reduced widths of the tiny test models, another stem type) runs the library ops of the module."""
        st = self.stem
        conv = st[0] if isinstance(st, nn.Sequential) and len(st) == 3 else None
        ok = (self.use_fused_stem and conv is not None and isinstance(conv, nn.Conv2d) and isinstance(st[1], nn.BatchNorm2d)
              and isinstance(st[2], nn.ReLU) and ops.get_math_mode() == "bf16x3" and conv.in_channels == 3
              and conv.out_channels in (32, 64) and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
              and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and (x.shape[3] - 1) // 2 + 1 <= 256)
        if not ok:
            return st(x)
        bn = st[1]
        src = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in src)
        if getattr(self, "_stem_key", None) != key:   # folded once, until a parameter or buffer changes
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                self._stem_frag = ops.pack_stem3_weights(conv.weight * scale.view(-1, 1, 1, 1))
                self._stem_shift = (bn.bias - bn.running_mean * scale).float().contiguous()
            self._stem_key = key
        return ops.from_nhwc(ops.stem3_conv(ops.as_nhwc(x), self._stem_frag, self._stem_shift, conv.out_channels, relu=True))

    def forward(self, x, temperature):
        _eval_only(self, x)
        in_shape = tuple(x.shape)
        x = x.contiguous(memory_format=torch.channels_last)
        x = self._stem_forward(x)
        stats = []
        sizes = [len(list(stage.children())) for stage in self.trunk_output.children()]
        prev = None
        for blk in self.blocks():
            blk.f.last_carry = None        # (a carry is only valid inside one forward)
        for j, blk in enumerate(self.blocks()):
            if self._tap is not None:     # debug tap (bench / tests): sees every block's input; off by default
                self._tap(j, blk.f, x)
            # layer skip: the images the previous block skipped are unchanged -> their channel sums are carried to this block's masker
            blk.f._carry_in = (prev.f.last_carry if (self.use_layer_carry and prev is not None and blk.proj is None and blk.stride == 1
                                                     and prev.proj is None and prev.stride == 1    # the producer leaves skipped images unchanged
                                                     and blk.f.dyn_mode == "spatial" and prev.f.dyn_mode == "spatial" and blk.f.mask_size == 1
                                                     and prev.f.mask_size == 1 and blk.f.forced_spatial_mask is None
                                                     and getattr(prev.f, "last_carry", None) is not None) else None)
            x, st = blk.run_dynamic(x, inplace=self.inplace_residual, defer_stats=True)
            prev = blk
            stats.append(st)
        if all(s.numel() == 3 for s in stats):      # spatial / layer-skip blocks only: channel sparsity 1, appended once
            st = torch.stack(stats)
            st = torch.cat((st, torch.ones(st.shape[0], 1, device=st.device)), dim=1)
        else:
            one = torch.ones(1, device=x.device)
            st = torch.stack([s if s.numel() == 4 else torch.cat((s, one)) for s in stats])
        s3, s2, s1, cs = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
        perc, flops = self.flops_from_sparsities(in_shape, s3, s2, s1, cs)
        x = self.avgpool(x)
        x = x.flatten(start_dim=1)
        x = self.fc(x)
        split = lambda v: list(torch.split(v, sizes))
        return x, split(s3), split(s2), split(s1), split(cs), perc, flops

    # ---- FLOPs bookkeeping (laud_regnet.py:179-203, 286-292, 576-611) from the input shape and the sparsities only
    def flops_table(self, x_shape):
        """(terms [n_blocks][6] = masker, a, b, c, proj, se per block; static FLOPs of stem, average pool, classifier)."""
        _, c_in, h, w = x_shape
        conv = self.stem[0]
        k, s, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        h, w = (h + 2 * pd - k) // s + 1, (w + 2 * pd - k) // s + 1
        c = conv.out_channels
        static = c_in * c * h * w * k * k
        terms = []
        for blk in self.blocks():
            terms.append(blk.flops_terms((1, c, h, w)))
            c = blk.f.c[0].out_channels
            h, w = (h - 1) // blk.stride + 1, (w - 1) // blk.stride + 1
        static += c + c * self.fc.out_features
        return terms, static

    def flops_from_sparsities(self, x_shape, s3, s2, s1, cs):
        """(flops_perc [n_blocks], flops) from per-block sparsities; see laud_resnet.ResNet.flops_from_sparsities."""
        flat = lambda v: torch.cat([t.reshape(-1) for t in v]) if isinstance(v, (list, tuple)) else v
        s3, s2, s1, cs = (flat(v).double() for v in (s3, s2, s1, cs))   # fp64 inside: the result does not depend on summation order
        key = (str(s3.device), tuple(x_shape[1:]))
        if getattr(self, "_terms_key", None) != key:
            terms, static = self.flops_table(x_shape)
            self._terms_key = key
            self._terms = torch.tensor(terms, dtype=torch.float64, device=s3.device)      # [n_blocks, 6]
            self._static_flops = float(static)
        tm = self._terms
        sparse = tm[:, 0] + tm[:, 1] * cs * s1
        sparse = sparse + tm[:, 2] * cs ** 2 * s2
        sparse = sparse + tm[:, 3] * cs * s3
        sparse = sparse + tm[:, 4]
        perc = sparse / tm[:, :5].sum(dim=1)
        flops = sparse.sum() + tm[:, 5].sum() + self._static_flops
        return perc.float(), flops.float()

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


def _lad_regnet(arch, block_params, pretrained, progress, **kwargs):
    if pretrained:
        raise LdnError("pretrained=True needs network access; load a state_dict explicitly")
    kwargs.pop("norm_layer", None)
    return LAD_RegNet(block_params, norm_layer=lambda c: nn.BatchNorm2d(c, eps=1e-05, momentum=0.1), **kwargs)


_Y = {"400mf": dict(depth=16, w_0=48, w_a=27.89, w_m=2.09, group_width=8),
      "800mf": dict(depth=14, w_0=56, w_a=38.84, w_m=2.4, group_width=16),
      "1_6gf": dict(depth=27, w_0=48, w_a=20.71, w_m=2.65, group_width=24),
      "3_2gf": dict(depth=21, w_0=80, w_a=42.63, w_m=2.66, group_width=24),
      "8gf": dict(depth=17, w_0=192, w_a=76.82, w_m=2.19, group_width=56),
      "16gf": dict(depth=18, w_0=200, w_a=106.23, w_m=2.48, group_width=112)}


def _factory(tag):
    def make(pretrained=False, progress=True, **kwargs):
        params = BlockParams.from_init_params(se_ratio=0.25, **_Y[tag])
        return _lad_regnet(f"regnet_y_{tag}", params, pretrained, progress, **kwargs)
    make.__name__ = f"lad_regnet_y_{tag}"
    make.__doc__ = f"laud_regnet.py: lad_regnet_y_{tag}"
    return make


lad_regnet_y_400mf = _factory("400mf")
lad_regnet_y_800mf = _factory("800mf")
lad_regnet_y_1_6gf = _factory("1_6gf")
lad_regnet_y_3_2gf = _factory("3_2gf")
lad_regnet_y_8gf = _factory("8gf")
lad_regnet_y_16gf = _factory("16gf")
