"""laudnet_amd -- MI355X-native implementation of LAUDNet's dynamic-inference hot path.

The package holds only what that path needs: `csrc/` (HIP kernels + the C ABI of include/ldn_hip.h),
`_lib.py`/`ops.py` (ctypes binding) and `laud_resnet.py` (host-side mirror of the reference's
nn.Module surface).  There is no CPU or PyTorch fallback for the hot path.
"""
import os as _os

# Kernel arguments in device memory (ROCm runtime switch HIP_FORCE_DEV_KERNARG, read when HIP initialises): the ~150-250 launches of a forward
# each start without an argument fetch over PCIe.  OPT-IN since round 6 (importing a library must not change the process environment behind its
# host's back): LDN_DEV_KERNARG=1 sets it here -- effective only when laudnet_amd is imported before the process's first HIP call -- and an
# explicit HIP_FORCE_DEV_KERNARG always wins.  bench.py sets the runtime switch itself, for every leg of its process alike.
if _os.environ.get("LDN_DEV_KERNARG", "0") not in ("", "0"):
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from ._lib import LdnError, load as load_library  # noqa: F401,E402
from .laud_resnet import (Bottleneck, ExpandMask, Masker_channel_conv_linear, Masker_channel_MLP,  # noqa: F401
                          Masker_spatial, ResNet, uni_resnet50, uni_resnet101)

from .laud_regnet import (LAD_RegNet, BlockParams, lad_regnet_y_400mf, lad_regnet_y_800mf,  # noqa: F401
                          lad_regnet_y_1_6gf, lad_regnet_y_3_2gf, lad_regnet_y_8gf, lad_regnet_y_16gf)

from . import sparsity_loss  # noqa: F401,E402  (the criteria the caller applies to the 7-tuple, train/main.py:311,670)
from . import training  # noqa: F401,E402  (forward + backward of spatial / layer blocks under frozen BatchNorm on the packed kernels)

__version__ = "0.1.0"
