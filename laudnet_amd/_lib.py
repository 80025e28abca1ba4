"""ctypes binding of libldn_hip.so (include/ldn_hip.h).  There is NO fallback: if the library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LDN_LIB_PATH", os.path.join(_PKG, "libldn_hip.so"))  # override: kernel tuning only

_P = C.c_void_p
_I = C.c_int

# name -> argtypes, in the order of include/ldn_hip.h
SIGNATURES = {
    "ldn_last_error": ([], C.c_char_p),
    "ldn_version": ([], _I),
    "ldn_debug_violations": ([C.POINTER(_I), C.POINTER(_I), _I], _I),
    "ldn_plan_timeouts": ([C.POINTER(_I), _I], _I),
    "ldn_fault_flag": ([], C.POINTER(_I)),
    "ldn_device_cus": ([C.POINTER(_I)], _I),
    "ldn_default_math_mode": ([], _I),
    "ldn_spatial_masker": ([_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "ldn_spatial_masker_workspace_bytes": ([_I, _I, _I, _I, _I], C.c_size_t),
    "ldn_mask_to_index": ([_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "ldn_mask_to_index_workspace_bytes": ([_I, _I, _I, _I], C.c_size_t),
    "ldn_mask_plan_fits": ([_I, _I, _I, _I, _I], _I),
    "ldn_hint_rows": ([_I], _I),
    "ldn_layer_index": ([_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "ldn_layer_head": ([_P, _I, _I, _I, _P, _P, _I, _P, _P, _P], _I),
    "ldn_mask_plan": ([_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "ldn_conv_rows_pool": ([_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P], _I),
    "ldn_conv_rows_ps": ([_P, _I, _I, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P], _I),
    "ldn_conv3x3_rows_ps": ([_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P], _I),
    "ldn_gather_rows": ([_P, _I, _P, _P, _I, _I, _P, _I, _P], _I),
    "ldn_scatter_add_relu": ([_P, _I, _P, _P, _I, _I, _P, _I, _P, _I, _P], _I),
    "ldn_conv_rows": ([_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P], _I),
    "ldn_conv_rows_split": ([_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _I, _I,
                             _I, _I, _P, _P, _P], _I),
    "ldn_conv_rows_f32": ([_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _I, _I,
                             _I, _I, _P, _P, _P], _I),
    "ldn_row_stats": ([_P, _I, _I, _I, C.c_float, _P, _P], _I),
    "ldn_row_stats_list": ([_P, _I, _I, _I, C.c_float, _P, _P, _P, _P], _I),
    "ldn_channel_masker_splits": ([_I], _I),
    "ldn_channel_masker": ([_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P], _I),
    "ldn_channel_masker_workspace_bytes": ([_I, _I, _I], C.c_size_t),
    "ldn_forward_stats": ([_P, _I, _P, _P, _I, _P, C.c_double, _I, _P, _P, _P, _P], _I),
    "ldn_conv_packed": ([_P, _I, _I, _P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P,
                         _I, _P, _I, _P, _P, _I, _P, _I, _I, _P], _I),
    "ldn_grouped_conv3x3_rows": ([_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _I, _P], _I),
    "ldn_grouped16_weight_bytes": ([_I], C.c_size_t),
    "ldn_grouped16_conv3x3_rows": ([_P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P], _I),
    "ldn_grouped16_images_fit": ([_I, _I, _I], _I),
    "ldn_grouped16_conv3x3_images": ([_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P], _I),
    "ldn_plan_work_zeroed": ([_I], _I),
    "ldn_coarsen_cell_means": ([_P, _I, _I, _I, _P, _P], _I),
    "ldn_grouped16_images_bands": ([_I, _I, _I, _I, _I], _I),
    "ldn_grouped16_conv3x3_images_gap": ([_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P], _I),
    "ldn_se_gate_slots": ([_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "ldn_conv_rows_gated": ([_P, _I, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P], _I),
    "ldn_grouped_conv3x3_image": ([_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P], _I),
    "ldn_se_packed": ([_P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P], _I),
    "ldn_se_packed_workspace_bytes": ([_I, _I, _I], C.c_size_t),
    "ldn_conv_image": ([_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _I,
                        _P, _I, _P, _I, _P, _I, _I, _P], _I),
    "ldn_bottleneck_head": ([_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P], _I),
    "ldn_bottleneck_head_f32": ([_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P], _I),
    "ldn_bottleneck_tail_splits": ([_I, _I, _I, _I], _I),
    "ldn_bottleneck_tail": ([_P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P], _I),
    "ldn_bottleneck_tail_f32": ([_P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P], _I),
    "ldn_x_split_bytes": ([C.c_size_t, _I], C.c_size_t),
    "ldn_bottleneck_head_split": ([_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P], _I),
    "ldn_bottleneck_tail_proj_fits": ([_I, _I, _I, _I], _I),
    "ldn_bottleneck_tail_proj": ([_P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P], _I),
    "ldn_stem_weight_bytes": ([_I], C.c_size_t),
    "ldn_stem_conv_pool": ([_P, _I, _I, _I, _P, _P, _I, _P, _I, _I, _P], _I),
    "ldn_stem_gap_splits": ([_I, _I], _I),
    "ldn_stem_conv_pool_gap": ([_P, _I, _I, _I, _P, _P, _I, _P, _I, _I, _P, _P], _I),
    "ldn_stem3_weight_bytes": ([_I], C.c_size_t),
    "ldn_stem3_conv": ([_P, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _P], _I),
    "ldn_packed_mha": ([_P, _I, _P, _P, _I, _I, _I, _I, C.c_float, _P, _I, _P], _I),
    "ldn_packed_mha_heads": ([_P, _I, _P, _P, _I, _I, _I, _I, C.c_float, _P, _P, _I, _P], _I),
    "ldn_bottleneck_chain_fits": ([_I, _I, _I, _I, _I, _I], _I),
    "ldn_bottleneck_smallmap_fits": ([_I, _I, _I, _I, _I], _I),
    "ldn_bottleneck_smallmap": ([_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P], _I),
    "ldn_bottleneck_chain": ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P], _I),
    "ldn_bottleneck_chain_f32": ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P], _I),
}

_lib = None


class LdnError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises LdnError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdnError(f"{LIB_PATH} not found: build it with `python -m laudnet_amd.build` "
                       "(laudnet_amd has no CPU or PyTorch fallback for the hot path)")
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr(t=None):
    """Launch stream = the current stream OF THE TENSORS' DEVICE (require_device has checked that it is also the current
    device, which is what the HIP runtime launches on)."""
    return C.c_void_p(torch.cuda.current_stream(t.device if t is not None else None).cuda_stream)


_SYNC_CALLS = os.environ.get("LDN_SYNC_CALLS", "0") != "0"    # debugging: synchronize after every library call and name it on stderr
_n_calls = 0


_fault = None          # the library's fault word (ldn_fault_flag), fetched with the first device tensor (require_device)


def _arm_fault_flag():
    global _fault
    if _fault is None:
        p = load().ldn_fault_flag()
        _fault = p if p else False


def check(status: int, what: str, fault_ok: bool = False):
    if status != 0:
        msg = load().ldn_last_error()
        raise LdnError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")
    if _fault and not fault_ok and _fault[0] != 0:
        # a kernel's bounded wait (one-launch list build, chained stage hand-off) ran into its bound: it left EMPTY lists / lost values
        raise LdnError(f"{what}: a device-side bounded wait ran into its time bound since the last reset -- results computed since are "
                       "invalid (ops.plan_timeouts() counts the events, ops.plan_timeouts(reset=True) re-arms)")
    if _SYNC_CALLS:
        global _n_calls
        import sys
        import torch
        _n_calls += 1
        print(f"[ldn] {_n_calls} {what} ...", end="", file=sys.stderr, flush=True)
        if torch.cuda.is_current_stream_capturing():
            print(" (capturing)", file=sys.stderr, flush=True)
            return
        torch.cuda.synchronize()
        print(" ok", file=sys.stderr, flush=True)


def require_device(*tensors):
    """Every tensor must live on ONE HIP device and that device must be the current one: libldn_hip.so launches on the
    runtime's current device and takes the stream from it -- a tensor on another GPU would be a foreign pointer there."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise LdnError("laudnet_amd ops need tensors on a HIP device (cuda:N); there is no CPU path")
        if cur is None:
            cur = torch.cuda.current_device()
            if _fault is None:
                _arm_fault_flag()
        if t.device.index != cur:
            raise LdnError(f"tensor on {t.device} but the current device is cuda:{cur}: wrap the call in "
                           f"torch.cuda.device({t.device.index}) (kernels are launched on the current device's stream)")
