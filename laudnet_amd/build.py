"""Build libldn_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libldn_hip.so")
SOURCES = ["ldn_conv_image.hip", "ldn_index.hip", "ldn_regnet.hip", "ldn_tail.hip", "ldn_dense.hip"]
HEADERS = [os.path.join(CSRC, "ldn_common.h"), os.path.join(os.path.dirname(PKG), "include", "ldn_hip.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


DEBUG_LIB = os.path.join(PKG, "libldn_hip_debug.so")


def build_debug(force: bool = False, verbose: bool = True) -> str:
    """The LDN_DEBUG build (device-side index-bounds checks, include/ldn_hip.h: ldn_debug_violations); select it with
    LDN_LIB_PATH=laudnet_amd/libldn_hip_debug.so."""
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    if not force and os.path.exists(DEBUG_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(DEBUG_LIB) for d in deps):
        return DEBUG_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-DLDN_DEBUG",
           "-o", DEBUG_LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[laudnet_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return DEBUG_LIB


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[laudnet_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--debug" in sys.argv or os.path.exists(DEBUG_LIB):     # an existing audit build is kept in step with the sources
        build_debug(force="--force" in sys.argv)
