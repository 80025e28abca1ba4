"""Reference-generated fixtures of LAD-RegNet with TWO spatial mask groups per block (spatial_mask_channel_group = 2:
laud_regnet.py:145-147,172-177,198 with models/utils.py:27-33,74-89) -> regnet_extra.pt: patch masks, layer skip (one mask per
image and channel group) and `both` mode, masker-produced and injected masks.

Kept apart from make_golden.py so that the committed regnet_tiny.pt stays byte-identical.  Run in the build container (imports the
reference from /root/reference with make_golden.py's torchvision-0.14 stand-ins; stores tensors only):
    python tests/golden/make_regnet_groups_golden.py
"""
import os

import torch

import make_golden as MG

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    MG.load_reference()              # registers the stub `models` package the RegNet file imports from
    G = MG.load_reference_regnet()
    MG.REGNET_CASES = {
        "spatial_g2_grp2": dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[2, 2, 2, 1], spatial_mask_channel_group=[2, 2, 2, 2]),
        "layerskip_grp2": dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[16, 8, 4, 2], spatial_mask_channel_group=[2, 2, 2, 2]),
        "both_grp2": dict(dyn_mode=["both"] * 4, mask_spatial_granularity=[4, 2, 2, 1], channel_dyn_granularity=[2, 2, 2, 2],
                          channel_masker=["MLP"] * 4, channel_masker_layers=[1, 1, 1, 1], spatial_mask_channel_group=[2, 2, 2, 2]),
    }
    full = MG.make_regnet(G)
    out = {"tiny_params": full["tiny_params"], "cases": full["cases"]}
    # make_regnet cuts the injected masks of a case NAMED "layerskip" to one bit per image; the two-group layer-skip case keeps its
    # [B, 2, 1, 1] masks because its name differs -- check the shapes the reference ran with
    torch.save(out, os.path.join(HERE, "regnet_extra.pt"))
    print("regnet_extra.pt", os.path.getsize(os.path.join(HERE, "regnet_extra.pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
