"""The detection backbone of the reference on the HIP path: mmdetection-2.21.0/mmdet/models/backbones/lad_mmdet_resnet.py
(LAD_MMDet_ResNet :332-751; the 3.3.0 fork carries the same class), eval mode.

Same constructor keywords, sub-module names (=> state_dict keys: conv1, bn1, layer{1-4}.{i}.conv{1,2,3} / bn{1,2,3} /
downsample.{0,1} / masker_channel.* | masker_spatial.*) and return value -- (tuple of the four stage outputs, `additional` dict,
`model_configs` dict) -- as the reference, without mmcv: the class is a plain nn.Module (register it with mmdet's BACKBONES
registry where mmdet is installed).  The blocks are laudnet_amd.laud_resnet.Bottleneck: inputs of any
H x W (multiples of 32, as mmdet pads them) run through the same kernels as the classifier -- the per-image channel-subset
convolutions (channel mode) or the packed rows of the kept images (layer mode: ldn_mask_to_index builds its lists by bands of rows
on maps that do not fit one workgroup's LDS); the masks follow the actual map size (lad_mmdet_resnet.py:274).

Not built (raise): deep_stem, avg_down, DCN, plugins, style='caffe', dilations != 1, gradient checkpointing, training mode
(frozen_stages / norm_eval only matter there and are accepted and recorded).
"""
from __future__ import annotations

import torch

from ._lib import LdnError
from .laud_resnet import Bottleneck, ResNet, _eval_only

ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}      # lad_mmdet_resnet.py:387-393


class LAD_MMDet_ResNet(ResNet):
    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch", deep_stem=False, avg_down=False,
                 frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True,
                 pretrained=None, init_cfg=None, sparsity_target=None, temperature_0=None, temperature_t=None,
                 spatial_mask_channel_group=[1, 1, 1, 1], mask_spatial_granularity=[1, 1, 1, 1],
                 channel_dyn_granularity=[1, 1, 1, 1], dyn_mode=["both", "both", "both", "both"],
                 channel_masker=["MLP", "MLP", "MLP", "MLP"], channel_masker_layers=[1, 1, 1, 1],
                 reduction_ratio=[16, 16, 16, 16]):
        if depth not in ARCH:
            raise KeyError(f"invalid depth {depth} for resnet")
        if (deep_stem or avg_down or dcn is not None or plugins is not None or style != "pytorch" or with_cp or conv_cfg is not None
                or tuple(dilations) != (1, 1, 1, 1) or num_stages != 4 or tuple(strides) != (1, 2, 2, 2) or in_channels != 3
                or (norm_cfg or {}).get("type", "BN") != "BN" or (stem_channels not in (None, base_channels))):
            raise LdnError("LAD_MMDet_ResNet on the HIP path: plain ResNet stem/stages only (no deep_stem / avg_down / DCN / plugins / "
                           "caffe style / dilation / non-BN norm)")
        if any(m not in ("channel", "layer") for m in dyn_mode):
            raise LdnError("LAD_MMDet_ResNet: the reference's detection block builds maskers for dyn_mode 'channel' and 'layer' only "
                           "(lad_mmdet_resnet.py:160-177)")
        if pretrained is not None:
            raise LdnError("pretrained checkpoints need network access; load a state_dict explicitly")
        # the block never receives reduction_ratio in the reference (lad_mmdet_resnet.py:522-529): its default 16 is used
        super().__init__(Bottleneck, list(ARCH[depth]), num_classes=1, zero_init_residual=zero_init_residual,
                         width_mult=base_channels / 64.0, input_size=224, spatial_mask_channel_group=spatial_mask_channel_group,
                         mask_spatial_granularity=mask_spatial_granularity, channel_dyn_granularity=channel_dyn_granularity,
                         dyn_mode=list(dyn_mode), channel_masker=channel_masker, channel_masker_layers=channel_masker_layers,
                         reduction_ratio=[16, 16, 16, 16])
        del self.fc, self.avgpool                      # a backbone: no classifier head (and no such state_dict keys)
        self.depth, self.out_indices = depth, tuple(out_indices)
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.sparsity_target = sparsity_target
        self.temperature_0, self.temperature_t = temperature_0, temperature_t
        self.feat_dim = Bottleneck.expansion * base_channels * 8

    def _head_flops(self, c):
        return 0

    def forward(self, x, iter_now=0, len_loader=100):
        """-> (outs, additional, model_configs) (lad_mmdet_resnet.py:680-751).  As in the reference, ALL four stage outputs are
        returned, whatever out_indices says (:711-730 append unconditionally)."""
        _eval_only(self, x)
        in_shape = tuple(x.shape)
        if in_shape[2] % 32 or in_shape[3] % 32:
            raise LdnError(f"LAD_MMDet_ResNet: input {in_shape[2]}x{in_shape[3]} must be a multiple of 32 (mmdet pads to size_divisor 32)")
        x = self._stem_forward(x)
        outs = []
        x, stats, sizes = self._run_blocks(x, stage_outs=outs, gap0=self.__dict__.pop("_stem_gap", None))
        st = self._stack_stats(stats, x.device)
        s3, s2, s1, cs = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
        perc, flops = self.flops_from_sparsities(in_shape, s3, s2, s1, cs)
        dense = (self._terms.sum() + self._static_flops).float()      # shape-only: every block's dense MACs + the stem's (:691-695)
        split = lambda v: list(torch.split(v, sizes))
        additional = {"spatial_sparsity_conv3": split(s3), "spatial_sparsity_conv2": split(s2), "spatial_sparsity_conv1": split(s1),
                      "channel_sparsity": split(cs), "flops_perc_list": perc, "flops": flops, "dense_flops": dense}
        return tuple(outs), additional, {"dyn_mode": self.dyn_mode, "sparsity_target": self.sparsity_target}
