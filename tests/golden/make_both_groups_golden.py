"""Reference-generated fixtures of a `both`-mode bottleneck with TWO spatial mask groups (spatial_mask_channel_group = 2,
models/utils.py:27-33,74-89 together with the channel mask of laud_resnet.py:101-103), strides 1 and 2 -> blocks_extra.pt.

Kept apart from make_golden.py so that the committed blocks_s{1,2}.pt stay byte-identical (make_golden.py walks its case table
with a running seed).  Run in the build container (imports the reference from /root/reference; stores tensors only):
    python tests/golden/make_both_groups_golden.py
"""
import os

import torch

import make_golden as MG

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    U, R = MG.load_reference()
    MG.BLOCK_CASES["both_grp2"] = ("both", 2, 2, 2)     # (dyn_mode, spatial granularity, channel granularity, spatial groups)
    out = {}
    for stride, seed in ((1, 900), (2, 910)):
        out[f"both_grp2_s{stride}"] = MG.make_block(U, R, "both_grp2", stride, seed)
    torch.save(out, os.path.join(HERE, "blocks_extra.pt"))
    print("blocks_extra.pt", os.path.getsize(os.path.join(HERE, "blocks_extra.pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
