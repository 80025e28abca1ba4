"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/ldn_hip.h declares; the Python mirror keeps the reference's state_dict surface; the product path
refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ldn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from laudnet_amd import build, _lib
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"libldn_hip.so does not export {n}"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES and include/ldn_hip.h disagree"
    assert _lib.load().ldn_version() >= 100


def test_argument_validation_without_gpu():
    from laudnet_amd import _lib
    lib = _lib.load()
    # NULL pointers / bad shapes are rejected before any launch
    assert lib.ldn_conv_rows(None, 0, None, 1, None, 10, None, 8, 8, None, None, 1, None, None, None, 0, None, 8, -1, None) == -1
    assert b"null" in lib.ldn_last_error()
    assert lib.ldn_gather_rows(None, 4, None, None, 1, 3, None, 4, None) == -1
    assert lib.ldn_channel_masker_splits(3136) >= 1


def test_math_mode_is_an_argument_not_a_global():
    """no global mutable state.  The library exports no setter; the mode travels as the math_mode argument of
    every conv call (validated before any launch); the host-side default is thread-local."""
    import threading
    from laudnet_amd import _lib, ops
    lib = _lib.load()
    assert not hasattr(lib, "ldn_set_math_mode") and not hasattr(lib, "ldn_get_math_mode")
    assert lib.ldn_default_math_mode() in (0, 1)
    # a bad mode is an argument error of the call itself (checked after the pointer checks: give it non-NULL dummies)
    dummy = ctypes.c_void_p(16)
    assert lib.ldn_conv_rows(dummy, 8, None, 1, None, 10, dummy, 8, 8, None, dummy, 1, None, None, None, 0, dummy, 8, 7, None) == -1
    assert b"math_mode" in lib.ldn_last_error()
    before = ops.get_math_mode()
    try:
        ops.set_math_mode("bf16x3")
        assert ops.get_math_mode() == "bf16x3" and ops._mm() == 1 and ops._mm("fp32") == 0
        seen = []
        t = threading.Thread(target=lambda: seen.append(ops._mm()))   # another thread does not inherit this thread's default
        t.start()
        t.join()
        assert seen == [-1]
        with pytest.raises(_lib.LdnError):
            ops.set_math_mode("tf32")
    finally:
        ops.set_math_mode(before if before != ops.get_math_mode() else None)
        ops.set_math_mode(None)


def test_workspace_twins_without_gpu():
    from laudnet_amd import _lib
    lib = _lib.load()
    splits = lib.ldn_channel_masker_splits(3136)
    assert lib.ldn_channel_masker_workspace_bytes(256, 3136, 256) == 256 * splits * 256 * 4
    assert lib.ldn_spatial_masker_workspace_bytes(4, 56, 56, 64, 14) == 4 * 14 * 14 * 64 * 4    # pooled to 14x14: the patches' pooled means (patch carry)
    assert lib.ldn_spatial_masker_workspace_bytes(4, 56, 56, 64, 56) == 0          # one logit per pixel: no scratch
    assert lib.ldn_spatial_masker_workspace_bytes(4, 56, 56, 64, 1) == 4 * splits * 64 * 4
    assert lib.ldn_mask_to_index_workspace_bytes(256, 14, 14, 1) == (3 * 256 + 4) * 4   # three counts per image (+ a spare word) for the one-launch build
    assert lib.ldn_mask_plan_fits(14, 14, 56, 56, 1) == 1 and lib.ldn_mask_plan_fits(25, 25, 200, 304, 1) == 0
    assert lib.ldn_mask_to_index_workspace_bytes(2, 200, 304, 1) > 3 * 2 * 4          # banded build: one entry per (image, band)
    assert lib.ldn_se_packed_workspace_bytes(8, 320, 196) == 8 * (lib.ldn_channel_masker_splits(196) + 1) * 320 * 4


@pytest.mark.parametrize("name", ["r50_spatial_g1", "r101_channel2222", "r101_spatial4421", "r101_layer", "r50_mixed"])
def test_state_dict_surface_matches_reference(name):
    import laudnet_amd
    fx = load_golden("full_tiny.pt")[name]
    model = getattr(laudnet_amd, fx["factory"])(**fx["kw"])
    assert list(model.state_dict().keys()) == fx["keys"]
    assert sum(p.numel() for p in model.parameters()) == fx["n_params"]
    pol = model.get_optim_policies()
    assert [g["name"] for g in pol] == ["backbone_params", "masker_params"]
    assert sum(p.numel() for g in pol for p in g["params"]) == fx["n_params"]


def test_no_cpu_fallback():
    import laudnet_amd
    fx = load_golden("full_tiny.pt")["r50_spatial_g1"]
    model = laudnet_amd.uni_resnet50(**fx["kw"]).eval()
    with pytest.raises(laudnet_amd.LdnError):
        model(torch.zeros(1, 3, 224, 224), 1.0)
    with pytest.raises(laudnet_amd.LdnError):
        model.train()(torch.zeros(1, 3, 224, 224), 1.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "laudnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_regnet_state_dict_surface_matches_reference():
    import laudnet_amd
    fxs = load_golden("regnet_tiny.pt")
    bp = laudnet_amd.BlockParams(**fxs["tiny_params"])
    for name, fx in fxs["cases"].items():
        model = laudnet_amd.LAD_RegNet(bp, **fx["kw"])
        assert list(model.state_dict().keys()) == fx["keys"], name
        assert sum(p.numel() for p in model.parameters()) == fx["n_params"]
    for name, want in fxs["params"].items():
        import laudnet_amd.laud_regnet as R
        got = R.BlockParams.from_init_params(se_ratio=0.25, **R._Y[name.replace("lad_regnet_y_", "")])
        assert (got.depths, got.widths, got.group_widths) == (want["depths"], want["widths"], want["group_widths"]), name


# ---- header <-> ctypes SIGNATURES <-> INTEGRATION.md stub, argument by argument ----------------------------------------------
def _header_prototypes():
    """name -> list of argument classes ('P' pointer, 'I' int, 'F' float, 'D' double, 'Z' size_t) parsed from include/ldn_hip.h."""
    text = open(os.path.join(ROOT, "include", "ldn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"\b(?:int|size_t|const\s+char\s*\*|const\s+int\s*\*)\s*(ldn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    kinds.append("P")
                elif re.match(r"(const )?float\b", a):
                    kinds.append("F")
                elif re.match(r"(const )?double\b", a):
                    kinds.append("D")
                elif re.match(r"(const )?size_t\b", a):
                    kinds.append("Z")
                elif re.match(r"(const )?(int|int32_t|unsigned)\b", a):
                    kinds.append("I")
                else:
                    raise AssertionError(f"{name}: cannot classify argument {a!r}")
        protos[name] = kinds
    return protos


def _kind(ct):
    if ct is ctypes.c_int:
        return "I"
    if ct is ctypes.c_float:
        return "F"
    if ct is ctypes.c_double:
        return "D"
    if ct is ctypes.c_size_t:
        return "Z"
    if ct is ctypes.c_void_p or ct is ctypes.c_char_p or (isinstance(ct, type) and issubclass(ct, ctypes._Pointer)):
        return "P"
    raise AssertionError(f"unexpected ctypes type {ct}")


def test_ctypes_signatures_match_header_argument_by_argument():
    from laudnet_amd import _lib
    protos = _header_prototypes()
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    for name, (argtypes, _) in _lib.SIGNATURES.items():
        got = [_kind(t) for t in argtypes]
        assert got == protos[name], f"{name}: ctypes {''.join(got)} vs header {''.join(protos[name])}"


def test_integration_md_stub_matches_header():
    """INTEGRATION.md section B shows a ctypes stub: every `_lib.<fn>.argtypes = [...]` line in it must have exactly the
    arguments include/ldn_hip.h declares (a stale stub passes the stream where math_mode is expected)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    protos = _header_prototypes()
    env = {"_P": ctypes.c_void_p, "_I": ctypes.c_int, "C": ctypes}
    found = 0
    for m in re.finditer(r"_lib\.(ldn_[a-z0-9_]+)\.argtypes\s*=\s*(\[[^\]]*\](?:\s*\+\s*\[[^\]]*\]\s*\*\s*\d+)?)", text):
        name, expr = m.group(1), m.group(2)
        argtypes = eval(expr, env)    # the document's own Python: lists of _P / _I / C.c_float
        got = [_kind(t) for t in argtypes]
        assert name in protos, f"INTEGRATION.md binds {name}, which the header does not declare"
        assert got == protos[name], f"INTEGRATION.md: {name} has {len(got)} arguments {''.join(got)}, header {len(protos[name])} {''.join(protos[name])}"
        found += 1
    assert found >= 5
