"""Build libldn_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libldn_hip.so")
SOURCES = ["ldn_conv_image.hip", "ldn_index.hip", "ldn_regnet.hip", "ldn_tail.hip", "ldn_dense.hip", "ldn_stem.hip", "ldn_attn.hip", "ldn_grouped.hip", "ldn_small.hip", "ldn_rows3.hip"]
HEADERS = [os.path.join(CSRC, "ldn_common.h"), os.path.join(os.path.dirname(PKG), "include", "ldn_hip.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


DEBUG_LIB = os.path.join(PKG, "libldn_hip_debug.so")
OBJ_DIR = os.path.join(PKG, "_obj")      # per-source objects (git-ignored): only the sources that changed are recompiled
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _compile_and_link(lib: str, tag: str, extra: list, force: bool, verbose: bool) -> str:
    """One object per translation unit (compiled in parallel, only when its source or a header is newer), then one link."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = HEADERS + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]      # (every internal header: ldn_mlp.h, ldn_se_head.h ...)
    hdr_t = max(os.path.getmtime(h) for h in hdrs if os.path.exists(h))
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{tag}.o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), hdr_t):
            cmd = [hipcc] + FLAGS + extra + ["-c", "-o", obj, sp]
            if verbose:
                print("[laudnet_amd.build]", " ".join(cmd), file=sys.stderr)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise subprocess.CalledProcessError(job.returncode, cmd)
    if jobs or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print("[laudnet_amd.build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return lib


def build_debug(force: bool = False, verbose: bool = True) -> str:
    """The LDN_DEBUG build (device-side index-bounds checks, include/ldn_hip.h: ldn_debug_violations); select it with
    LDN_LIB_PATH=laudnet_amd/libldn_hip_debug.so."""
    return _compile_and_link(DEBUG_LIB, "debug", ["-DLDN_DEBUG"], force, verbose)


def build(force: bool = False, verbose: bool = True) -> str:
    return _compile_and_link(LIB, "rel", [], force, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--debug" in sys.argv or os.path.exists(DEBUG_LIB):     # an existing audit build is kept in step with the sources
        build_debug(force="--force" in sys.argv)
