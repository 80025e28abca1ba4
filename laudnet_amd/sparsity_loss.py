"""The sparsity criteria that consume the model's 7-tuple (host side of the caller contract).

The reference's eval loop calls `sparsity_criterion(epoch, flops_perc_list, flops)` on the tuple the model returns
(`train/main.py:636,670`, criterion built at `train/main.py:311`); the training loop does the same at `train/main.py:562-569`.
This module restates `utils/sparsity_loss_unify.py` (class names, constructor arguments and call signatures kept, so the
reference's harness can import it instead) as three shared terms over tensors -- no Python loop over blocks, so the criterion
runs as a handful of small device ops on the sparsity vectors the HIP path already keeps on the GPU, with no `.item()` sync:

  schedule(epoch)      progress = cos^2(clip(epoch / (0.33 * num_epochs), 0, 1) * pi/2)      (sparsity_loss_unify.py:15-16)
  band(v, target)      mean_i [ relu(v_i - upper)^2 + relu(lower - v_i)^2 ],  upper = top - progress * (top - target),
                       lower = progress * target                                            (sparsity_loss_unify.py:17-26)
  overall(flops)       (flops / full_flops - target)^2                                      (sparsity_loss_unify.py:27)

Everything is differentiable (relu / square), so the criteria are usable as a training loss by whoever supplies a training-mode
forward; the HIP path itself is inference only.  Floating point: sums over blocks are tree sums here and running sums
in the reference -- parity is asserted to 1e-6 relative (tests/test_sparsity_loss.py).
"""
from __future__ import annotations

import math
from typing import Sequence, Union

import torch
import torch.nn as nn

Values = Union[torch.Tensor, Sequence]


def _vec(values: Values) -> torch.Tensor:
    """A per-block list (floats, 0-dim tensors) or tensor -> 1-D tensor, kept on its device, graph preserved."""
    if isinstance(values, torch.Tensor):
        return values.reshape(-1)
    items = [v if isinstance(v, torch.Tensor) else torch.tensor(float(v)) for v in values]
    dev = next((v.device for v in items if v.is_cuda), items[0].device)
    return torch.stack([v.reshape(()).to(dev) for v in items])


class _Scheduled(nn.Module):
    """Shared pieces: the cosine schedule of the bounds, the band penalty and the overall-FLOPs term."""

    def __init__(self, target: float, num_epochs: int, full_flops: float):
        super().__init__()
        self.num_epochs = num_epochs
        self.full_flops = full_flops
        self._target = target

    def schedule(self, epoch: float) -> float:
        p = min(max(epoch / (0.33 * self.num_epochs), 0), 1)
        return math.cos(p * (math.pi / 2)) ** 2

    def band(self, values: Values, target: float, epoch: float, top: float = 1.0) -> torch.Tensor:
        v = _vec(values)
        progress = self.schedule(epoch)
        upper = top - progress * (top - target)
        lower = progress * target
        return (torch.relu(v - upper) ** 2 + torch.relu(lower - v) ** 2).sum() / v.numel()

    def overall(self, flops) -> torch.Tensor:
        return (flops / self.full_flops - self._target) ** 2


class SparsityCriterion_bounds(_Scheduled):
    """`utils/sparsity_loss_unify.py:6-29` -- the criterion `train/main.py:311` builds."""

    def __init__(self, sparsity_target, num_epochs, full_flops):
        super().__init__(sparsity_target, num_epochs, full_flops)
        self.sparsity_target = sparsity_target

    def forward(self, epoch, sparsity_list, flops):
        return self.band(sparsity_list, self.sparsity_target, epoch) + self.overall(flops)


class _ChannelAware(_Scheduled):
    def __init__(self, flops_perc_target=1.0, num_epochs=100, full_flops=4.1, channel_target=None):
        super().__init__(flops_perc_target, num_epochs, full_flops)
        self.flops_perc_target = flops_perc_target
        self.channel_target = math.sqrt(flops_perc_target) if channel_target is None else channel_target

    def _flops_terms(self, epoch, flops_perc_list, flops):
        return self.band(flops_perc_list, self.flops_perc_target, epoch) + self.overall(flops)


class SparsityCriterion(_ChannelAware):
    """`sparsity_loss_unify.py:31-69`: channel densities pulled to sqrt(target), FLOPs ratios kept in the band."""

    def __init__(self, flops_perc_target, num_epochs, full_flops):
        super().__init__(flops_perc_target, num_epochs, full_flops)

    def forward(self, epoch, channel_sparsity_list, flops_perc_list, flops):
        pull = torch.mean((channel_sparsity_list - self.channel_target) ** 2)
        return pull + self._flops_terms(epoch, flops_perc_list, flops)


class _PerStage(_ChannelAware):
    def __init__(self, flops_perc_target=1.0, num_epochs=100, full_flops=4.1, factor=1.0, channel_target=None,
                 dyn_mode=("both", "both", "both", "both")):
        super().__init__(flops_perc_target, num_epochs, full_flops, channel_target)
        self.spatial_target = flops_perc_target
        self.dyn_mode = list(dyn_mode)

    def _stages(self, *modes):
        return [i for i in range(4) if self.dyn_mode[i] in modes]


class SparsityCriterion_channel_factor(_PerStage):
    """`sparsity_loss_unify.py:71-106`: per-stage channel pull on the stages in `both` mode, weighted."""

    def __init__(self, flops_perc_target=1.0, num_epochs=100, full_flops=4.1, channel_loss_factor=1.0, channel_target=None,
                 dyn_mode=("both", "both", "both", "both")):
        super().__init__(flops_perc_target, num_epochs, full_flops, channel_loss_factor, channel_target, dyn_mode)
        self.channel_loss_factor = channel_loss_factor

    def forward(self, epoch, channel_sparsity_list, flops_perc_list, flops):
        pull = 0.0
        for i in self._stages("both"):
            pull = pull + torch.mean((channel_sparsity_list[i] - self.channel_target) ** 2)
        return self.channel_loss_factor * pull + self._flops_terms(epoch, flops_perc_list, flops)


class SparsityCriterion_cs(_PerStage):
    """`sparsity_loss_unify.py:108-150`: per-stage channel AND spatial pulls on the `both` stages."""

    def __init__(self, flops_perc_target=1.0, num_epochs=100, full_flops=4.1, cs_loss_factor=1.0, channel_target=None,
                 dyn_mode=("both", "both", "both", "both")):
        super().__init__(flops_perc_target, num_epochs, full_flops, cs_loss_factor, channel_target, dyn_mode)
        self.cs_loss_factor = cs_loss_factor

    def forward(self, epoch, channel_sparsity_list, spatial_sparsity_list, flops_perc_list, flops):
        pull = 0.0
        for i in self._stages("both"):
            pull = pull + torch.mean((channel_sparsity_list[i] - self.channel_target) ** 2)
            pull = pull + torch.mean((spatial_sparsity_list[i] - self.spatial_target) ** 2)
        return self.cs_loss_factor * pull + self._flops_terms(epoch, flops_perc_list, flops)


class SparsityCriterion_cs_v2(SparsityCriterion_cs):
    """`sparsity_loss_unify.py:152-198`: the pulls act on the MEAN density over all channel- (spatial-) dynamic stages."""

    def forward(self, epoch, channel_sparsity_list, spatial_sparsity_list, flops_perc_list, flops):
        dens_c = torch.cat([channel_sparsity_list[i] for i in self._stages("channel", "both")])
        dens_s = torch.cat([spatial_sparsity_list[i] for i in self._stages("spatial", "both")])
        pull = (torch.mean(dens_c) - self.channel_target) ** 2 + (torch.mean(dens_s) - self.spatial_target) ** 2
        return self.cs_loss_factor * pull + self._flops_terms(epoch, flops_perc_list, flops)


class SparsityCriterion_channel_bounds(_ChannelAware):
    """`sparsity_loss_unify.py:200-241`: the channel densities get a band of their own (top 1.0)."""

    channel_top = 1.0

    def __init__(self, flops_perc_target=1.0, num_epochs=100, full_flops=4.1, channel_loss_factor=1.0):
        super().__init__(flops_perc_target, num_epochs, full_flops)
        self.channel_loss_factor = channel_loss_factor

    def forward(self, epoch, channel_sparsity_list, flops_perc_list, flops):
        # the reference walks both lists with the FLOPs list's length (sparsity_loss_unify.py:224-231)
        n = len(flops_perc_list)
        chan = self.band(_vec(channel_sparsity_list)[:n], self.channel_target, epoch, top=self.channel_top)
        return self.channel_loss_factor * chan + self._flops_terms(epoch, flops_perc_list, flops)


class SparsityCriterion_channel_bounds_v2(SparsityCriterion_channel_bounds):
    """`sparsity_loss_unify.py:244-289`: the channel band starts from 0.85 instead of 1."""

    channel_top = 0.85
