"""Dense-emulation oracle for the LAUD-ResNet dynamic bottleneck (torch, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

This is a restatement, not a copy, of the reference algorithm.  Each piece cites
the reference lines (relative to /root/reference/imagenet_classification/) whose
behaviour it follows.  Attribute names of sub-modules match the reference so the
reference's state_dict loads with ``strict=True`` -- that is how the golden
fixtures (tests/golden/*.pt) are replayed through this file.

Extra, default-off hooks (not in the reference): every block has
``forced_spatial_mask`` / ``forced_channel_mask``; when set, the masker's
arithmetic is skipped and the given {0,1} mask is used instead ("identical
inputs/masks" parity runs).  Mask FLOPs bookkeeping is unchanged by the hook.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- L1
def broadcast_channel_mask(mask: torch.Tensor, channels: int) -> torch.Tensor:
    """[B,G] group mask -> [B,C,1,1]; group g owns channels [g*C/G,(g+1)*C/G).
    Follows models/utils.py:18-25 (apply_channel_mask)."""
    b, g = mask.shape
    if g not in (1, channels):
        mask = mask.repeat_interleave(channels // g, dim=1)
    return mask.reshape(b, -1, 1, 1)


def broadcast_spatial_mask(mask: torch.Tensor, channels: int) -> torch.Tensor:
    """[B,g,H,W] -> broadcastable over C; group g owns a contiguous C/g block.
    Follows models/utils.py:27-33 (apply_spatial_mask)."""
    g = mask.shape[1]
    if g not in (1, channels):
        mask = mask.repeat_interleave(channels // g, dim=1)
    return mask


def hard_decision(logits2: torch.Tensor, training: bool, temperature: float) -> torch.Tensor:
    """logits2: [B,2,...].  Eval: keep iff logit_keep >= logit_drop (ties keep),
    models/utils.py:60,127,165.  Train: Gumbel-softmax hard sample, :57,124,162."""
    if training:
        return F.gumbel_softmax(logits2, dim=1, tau=temperature, hard=True)[:, 0]
    return (logits2[:, 0] >= logits2[:, 1]).to(torch.float32)


class SpatialMaskerRef(nn.Module):
    """models/utils.py:35-65 (Masker_spatial)."""

    def __init__(self, in_channels: int, groups: int, mask_size: int):
        super().__init__()
        self.groups = groups
        self.mask_size = mask_size
        self.conv = nn.Conv2d(in_channels, 2 * groups, kernel_size=1, bias=True)
        # utils.py:41  (2g*Cin + Cin) per mask pixel
        self.flops_per_pixel = 2 * groups * in_channels + in_channels
        with torch.no_grad():  # utils.py:42-43 (bias[g] keeps its random init)
            self.conv.bias[:groups] = 5.0
            self.conv.bias[groups + 1:] = 0.0

    def flops_for(self, x: torch.Tensor) -> int:
        s = self.mask_size if self.mask_size < x.shape[2] else x.shape[2]
        sw = self.mask_size if self.mask_size < x.shape[2] else x.shape[3]
        return x.shape[1] * s * sw + self.flops_per_pixel * s * sw

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        pooled = F.adaptive_avg_pool2d(x, self.mask_size) if self.mask_size < x.shape[2] else x
        z = self.conv(pooled)
        b, c2, h, w = z.shape
        return z.view(b, 2, c2 // 2, h, w)

    def forward(self, x, temperature):
        mask = hard_decision(self.logits(x), self.training, temperature)
        return mask, mask.mean(), self.flops_for(x)


class ChannelMaskerMLPRef(nn.Module):
    """models/utils.py:92-131 (Masker_channel_MLP)."""

    def __init__(self, in_channels: int, groups: int, layers: int = 2, reduction: int = 16):
        super().__init__()
        if layers not in (1, 2):
            raise AssertionError("layers must be 1 or 2")
        self.groups = groups
        hidden = max(groups // reduction, 16)
        if layers == 2:
            self.conv = nn.Sequential(nn.Linear(in_channels, hidden), nn.ReLU(),
                                      nn.Linear(hidden, 2 * groups))
            self.mlp_flops = in_channels * hidden + hidden * 2 * groups
            last = self.conv[-1]
        else:
            self.conv = nn.Linear(in_channels, 2 * groups)
            self.mlp_flops = in_channels * 2 * groups
            last = self.conv
        with torch.no_grad():  # utils.py:106-111
            last.bias[:groups] = 2.0
            last.bias[groups + 1:] = -2.0

    def flops_for(self, x):
        return x.shape[1] * x.shape[2] * x.shape[3] + self.mlp_flops

    def logits(self, x):
        b, c = x.shape[:2]
        z = self.conv(F.adaptive_avg_pool2d(x, 1).view(b, c))
        return z.view(b, 2, z.shape[1] // 2)

    def forward(self, x, temperature):
        mask = hard_decision(self.logits(x), self.training, temperature)
        return mask, mask.mean(), self.flops_for(x)


class ChannelMaskerConvLinearRef(nn.Module):
    """models/utils.py:133-169 (Masker_channel_conv_linear)."""

    def __init__(self, in_channels: int, groups: int, reduction: int = 16):
        super().__init__()
        self.groups = groups
        mid = in_channels // reduction
        self.conv = nn.Sequential(nn.Conv2d(in_channels, mid, 1, bias=False),
                                  nn.BatchNorm2d(mid), nn.ReLU())
        self.linear = nn.Linear(mid, 2 * groups)
        with torch.no_grad():  # utils.py:145-146
            self.linear.bias[:groups] = 2.0
            self.linear.bias[groups + 1:] = -2.0
        self.head_flops = in_channels * in_channels // reduction + mid * 2 * groups  # :148
        self.mid = mid

    def flops_for(self, x):
        return self.mid * x.shape[2] * x.shape[3] + self.head_flops  # :153,157

    def logits(self, x):
        y = self.conv(x)
        b, c = y.shape[:2]
        z = self.linear(F.adaptive_avg_pool2d(y, 1).view(b, c))
        return z.view(b, 2, z.shape[1] // 2)

    def forward(self, x, temperature):
        mask = hard_decision(self.logits(x), self.training, temperature)
        return mask, mask.mean(), self.flops_for(x)


def expand_mask(mask: torch.Tensor, stride: int, padding: int) -> torch.Tensor:
    """models/utils.py:67-89 (ExpandMask.forward), restated without convolutions.

    zero-insert upsample by `stride` (value lands at [i*s, j*s]), OR across the
    mask's channel groups (the reference's ones-kernel is [g,g,k,k], utils.py:81),
    then a (2p+1)^2 box dilation with zero padding; returns bool [B,g,H*s,W*s]."""
    b, g, h, w = mask.shape
    m = mask.to(torch.float32)
    if stride > 1:
        up = m.new_zeros(b, g, h * stride, w * stride)
        up[:, :, ::stride, ::stride] = m
        m = up
    any_g = m.sum(dim=1, keepdim=True)
    k = 2 * padding + 1
    # box-sum == ones-conv; max-pool would give the same >0.5 decision
    boxed = F.avg_pool2d(any_g, k, stride=1, padding=padding, count_include_pad=True,
                         divisor_override=1) if k > 1 else any_g
    return (boxed > 0.5).expand(b, g, -1, -1)


# --------------------------------------------------------------------------- L2
class BottleneckRef(nn.Module):
    """models/laud_resnet.py:24-165 (Bottleneck), dense emulation."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, group_width=1, dilation=1,
                 spatial_mask_channel_group=1, channel_dyn_granularity=1, output_size=56,
                 mask_spatial_granularity=1, dyn_mode="both", channel_masker="conv_linear",
                 channel_masker_layers=2, reduction=16):
        super().__init__()
        assert dyn_mode in ("channel", "spatial", "both", "layer")
        assert channel_masker in ("conv_linear", "MLP")
        self.dyn_mode = dyn_mode
        width = int(planes * 1.0) * group_width          # laud_resnet.py:48-49
        assert channel_dyn_granularity <= width
        groups_c = width // channel_dyn_granularity      # :52
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=dilation, groups=group_width,
                               dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride
        self.macs_pp = (inplanes * width, width * width * 9 // group_width, width * planes * 4)  # :63-65
        self.ds_macs_pp = inplanes * planes * 4 if downsample is not None else 0             # :68
        self.output_size = output_size
        self.mask_size = output_size // mask_spatial_granularity if dyn_mode != "layer" else 1  # :72
        self.masker_spatial = None
        self.masker_channel = None
        if dyn_mode in ("spatial", "layer", "both"):
            self.masker_spatial = SpatialMaskerRef(inplanes, spatial_mask_channel_group, self.mask_size)
        if dyn_mode in ("channel", "both"):
            if channel_masker == "conv_linear":
                self.masker_channel = ChannelMaskerConvLinearRef(inplanes, groups_c, reduction)
            else:
                self.masker_channel = ChannelMaskerMLPRef(inplanes, groups_c, channel_masker_layers, reduction)
        self.forced_spatial_mask = None
        self.forced_channel_mask = None

    # -- mask production (with the parity hook)
    def _channel_mask(self, x, temperature):
        if self.forced_channel_mask is not None:
            m = self.forced_channel_mask.to(x.dtype)
            return m, m.mean(), self.masker_channel.flops_for(x)
        return self.masker_channel(x, temperature)

    def _spatial_mask(self, x, temperature):
        if self.forced_spatial_mask is not None:
            m = self.forced_spatial_mask.to(x.dtype)
            return m, m.mean(), self.masker_spatial.flops_for(x)
        return self.masker_spatial(x, temperature)

    def forward(self, state, temperature=1.0):
        x, s3_list, s2_list, s1_list, cs_list, perc_list, flops = state
        one = lambda: torch.tensor(1.0, device=x.device)
        use_c = self.dyn_mode in ("channel", "both")
        use_s = self.dyn_mode != "channel"

        c_flops = s_flops = 0
        if use_c:
            cmask, cs, c_flops = self._channel_mask(x, temperature)
        else:
            cs = one()                                                       # :99
        if use_s:
            m3, s3, s_flops = self._spatial_mask(x, temperature)
            m3 = F.interpolate(m3, size=self.output_size, mode="nearest")    # :106
            m2 = expand_mask(m3, 1, 0)                                       # :107
            s2 = m2.float().mean()
            m1 = expand_mask(m2, self.stride, 1)                             # :109
            s1 = m1.float().mean()
        else:
            s1, s2, s3 = one(), one(), one()                                 # :95

        sparse = c_flops + s_flops                                           # :112-113
        dense = c_flops + s_flops

        h = self.conv1(x)
        if use_c:
            h = h * broadcast_channel_mask(cmask, h.shape[1])                # pre-BN mask, :116-117
        h = F.relu(self.bn1(h))
        px = h.shape[2] * h.shape[3]
        dense += self.macs_pp[0] * px
        sparse = sparse + self.macs_pp[0] * px * cs * s1                     # :121

        h = self.conv2(h)
        if use_c:
            h = h * broadcast_channel_mask(cmask, h.shape[1])                # :124
        h = F.relu(self.bn2(h))
        px = h.shape[2] * h.shape[3]
        dense += self.macs_pp[1] * px
        sparse = sparse + self.macs_pp[1] * px * cs ** 2 * s2                # :129

        h = self.bn3(self.conv3(h))
        if use_s:
            h = h * broadcast_spatial_mask(m3, h.shape[1])                   # :133
        dense += self.macs_pp[2] * px
        sparse = sparse + self.macs_pp[2] * px * cs * s3                     # :136

        identity = x
        if self.downsample is not None:
            identity = self.downsample(x)
            ipx = identity.shape[2] * identity.shape[3]
            dense += self.ds_macs_pp * ipx
            sparse = sparse + self.ds_macs_pp * ipx                          # :140-141
        out = F.relu(h + identity)

        flops = flops + sparse
        perc = sparse / dense

        def push(lst, v):
            v = v.reshape(1)
            return v if lst is None else torch.cat((lst, v))

        return (out, push(s3_list, s3), push(s2_list, s2), push(s1_list, s1),
                push(cs_list, cs), push(perc_list, perc), flops)


# --------------------------------------------------------------------------- L3
class ResNetRef(nn.Module):
    """models/laud_resnet.py:167-363 (ResNet)."""

    def __init__(self, layers, num_classes=1000, width_mult=1.0, input_size=224,
                 spatial_mask_channel_group=(1, 1, 1, 1), mask_spatial_granularity=(1, 1, 1, 1),
                 channel_dyn_granularity=(1, 1, 1, 1), dyn_mode=("both",) * 4,
                 channel_masker=("MLP",) * 4, channel_masker_layers=(1, 1, 1, 1),
                 reduction_ratio=(16, 16, 16, 16), lr_mult=1.0, **_ignored):
        super().__init__()
        self.dyn_mode = list(dyn_mode)
        self.lr_mult = lr_mult
        self.inplanes = int(64 * width_mult)
        self.conv1 = nn.Conv2d(3, self.inplanes, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        for i, (mult, stride, down) in enumerate(((64, 1, 4), (128, 2, 8), (256, 2, 16), (512, 2, 32))):
            stage = self._stage(int(mult * width_mult), layers[i], stride, dict(
                output_size=input_size // down,
                spatial_mask_channel_group=spatial_mask_channel_group[i],
                mask_spatial_granularity=mask_spatial_granularity[i],
                channel_dyn_granularity=channel_dyn_granularity[i],
                dyn_mode=dyn_mode[i], channel_masker=channel_masker[i],
                channel_masker_layers=channel_masker_layers[i], reduction=reduction_ratio[i]))
            setattr(self, f"layer{i + 1}", stage)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(int(512 * width_mult) * 4, num_classes)
        for name, m in self.named_modules():                                 # :255-260
            if isinstance(m, nn.Conv2d) and "masker" not in name:
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _stage(self, planes, blocks, stride, kw):
        down = None
        if stride != 1 or self.inplanes != planes * 4:                       # :282-286
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        mods = [BottleneckRef(self.inplanes, planes, stride=stride, downsample=down, **kw)]
        self.inplanes = planes * 4
        mods += [BottleneckRef(self.inplanes, planes, **kw) for _ in range(1, blocks)]
        return nn.ModuleList(mods)

    def blocks(self):
        for i in range(4):
            for j, blk in enumerate(getattr(self, f"layer{i + 1}")):
                yield f"layer{i + 1}.{j}", blk

    def forward(self, x, temperature):
        cin = x.shape[1]
        x = F.relu(self.bn1(self.conv1(x)))
        flops = cin * x.shape[1] * x.shape[2] * x.shape[3] * 49              # :321
        x = self.maxpool(x)
        flops += x.shape[1] * x.shape[2] * x.shape[3] * 9                    # :324
        perc, per_stage = None, []
        for i in range(4):                                                   # :329-347
            state = (x, None, None, None, None, perc, flops)
            for blk in getattr(self, f"layer{i + 1}"):
                state = blk(state, temperature)
            x, s3, s2, s1, cs, perc, flops = state
            per_stage.append((s3, s2, s1, cs))
        x = self.avgpool(x)
        flops = flops + x.shape[1] * x.shape[2] * x.shape[3]                 # :350
        x = torch.flatten(x, 1)
        flops = flops + x.shape[1] * self.fc.out_features                    # :356
        logits = self.fc(x)
        cols = list(zip(*per_stage))
        return (logits, list(cols[0]), list(cols[1]), list(cols[2]), list(cols[3]), perc, flops)


def resnet50_ref(**kw):
    return ResNetRef([3, 4, 6, 3], **kw)


def resnet101_ref(**kw):
    return ResNetRef([3, 4, 23, 3], **kw)


def randomize_bn_(model: nn.Module, seed: int = 1) -> None:
    """Give every BatchNorm non-trivial eval statistics: fresh BN is
    identity-like and hides the pre-BN channel-mask constants (laud_resnet.py:116-117)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)


def randomize_maskers_(model: nn.Module, seed: int = 3) -> None:
    """Recipe (A) of SURVEY 8c: zero the last-layer bias of every masker and draw
    its weights from N(0,1) so that masker-produced masks are ~50 % dense."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if "masker" not in name:
                continue
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g))
                if m.bias is not None:
                    m.bias.zero_()
