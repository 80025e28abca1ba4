import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_generate_tests(metafunc):
    # GPU parity tests run once per arithmetic mode of the MFMA convolutions (include/ldn_hip.h: ldn_set_math_mode)
    if "math_mode" in metafunc.fixturenames:
        metafunc.parametrize("math_mode", ["fp32", "bf16x3"])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _no_list_build_timeouts():
    """At the end of a GPU session: no launch of the one-launch list build (k_plan) may have run into its time bound
    (ldn_plan_timeouts; such a launch leaves empty lists and would make a parity test fail for the wrong reason)."""
    yield
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from laudnet_amd import ops
        n = ops.plan_timeouts()
    except Exception:      # library not built / no device: nothing to check
        return
    assert n == 0, f"{n} list-build launches ran into their time bound during this session"
    # ... and, when the session ran on the LDN_DEBUG build (LDN_LIB_PATH=laudnet_amd/libldn_hip_debug.so pytest -m gpu), no device-side index audit
    # may have counted a violation (the release build reports -1: checks compiled away).  tests/test_hip_debug.py corrupts a list on purpose in
    # its own subprocess; deselect it for such a session.
    try:
        import ctypes
        from laudnet_amd import _lib
        lib = _lib.load()
        c, code = ctypes.c_int(0), ctypes.c_int(0)
        if lib.ldn_debug_violations(ctypes.byref(c), ctypes.byref(code), 0) != 0:
            return
    except Exception:
        return
    assert c.value <= 0, f"{c.value} device-side index-audit violations during this session (first code {code.value})"
