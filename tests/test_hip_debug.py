"""LDN_DEBUG build: the same forward passes through libldn_hip_debug.so, whose kernels audit every index list on the
device.  A clean run must report zero violations with results identical to the release build's; a deliberately corrupted
channel list must be counted (and must not trap).  Runs in a subprocess because the library path is fixed at import time."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, json, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests/golden")
import torch
import laudnet_amd
from laudnet_amd import _lib, ops
from fill import fill_state_dict, seeded_randn, seeded_bernoulli
lib = _lib.load()
def violations(reset=1):
    c, code = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.ldn_debug_violations(ctypes.byref(c), ctypes.byref(code), reset) == 0
    return c.value, code.value
res = {"initial": violations()}
logits = {}
for math in ("fp32", "bf16x3"):
    ops.set_math_mode(math)
    for mode, extra in (("channel", dict(channel_dyn_granularity=[2] * 4, channel_masker_layers=[2] * 4)),
                        ("spatial", dict(mask_spatial_granularity=[4, 4, 2, 1])),
                        ("both", dict(channel_dyn_granularity=[2] * 4, channel_masker_layers=[2] * 4, mask_spatial_granularity=[4, 4, 2, 1]))):
        kw = dict(dyn_mode=[mode] * 4, width_mult=0.5, input_size=128, num_classes=10, **extra)
        m = laudnet_amd.uni_resnet50(**kw).eval()
        m.load_state_dict(fill_state_dict(m.state_dict(), 5))
        m = m.cuda()
        blocks = [b for s in (1, 2, 3, 4) for b in getattr(m, "layer%%d" %% s)]
        for i, b in enumerate(blocks):
            if b.masker_spatial is not None:
                b.forced_spatial_mask = seeded_bernoulli((4, 1, b.masker_spatial.mask_size, b.masker_spatial.mask_size), 0.5, 100 + i)
            if b.masker_channel is not None:
                b.forced_channel_mask = seeded_bernoulli((4, b.masker_channel.channel_dyn_group), 0.62, 200 + i).cuda()
        with torch.no_grad():
            y = m(seeded_randn((4, 3, 128, 128), 9).cuda(), 1.0)[0]
        logits[mode + "/" + math] = y.double().abs().sum().item()
    rg = laudnet_amd.lad_regnet_y_400mf(dyn_mode=["channel"] * 4, channel_dyn_granularity=[2] * 4, channel_masker=["MLP"] * 4,
                                        channel_masker_layers=[2] * 4, num_classes=10, input_size=64).eval()
    rg.load_state_dict(fill_state_dict(rg.state_dict(), 6))
    rg = rg.cuda()
    with torch.no_grad():
        logits["regnet/" + math] = rg(seeded_randn((2, 3, 64, 64), 3).cuda(), 1.0)[0].double().abs().sum().item()
res["clean"] = violations()
res["logits"] = logits
# a corrupted channel list: an odd first entry breaks the aligned-pair invariant of the fused tail (reads stay inside the tensors)
ops.set_math_mode("bf16x3")
B, H, W = 2, 14, 64
idx = torch.arange(W, dtype=torch.int32).repeat(B, 1).cuda()
cnt = torch.full((B,), 32, dtype=torch.int32).cuda()
idx[0, 0] = 1
h1 = torch.zeros(B, H, H, W).cuda()
out = torch.zeros(B, H, H, 256).cuda()
ops.bottleneck_tail(h1, ops.pack_w2_pairs(torch.zeros(W, W, 3, 3).cuda()), ops.pack_w3_pairs(torch.zeros(256, W).cuda()), idx, cnt,
                    torch.ones(W).cuda(), torch.zeros(16, W).cuda(), torch.zeros(W).cuda(), torch.zeros(256).cuda(), out)
res["corrupted"] = violations()
print("RESULT " + json.dumps(res))
"""


def _run(lib):
    env = dict(os.environ, LDN_LIB_PATH=os.path.join(ROOT, "laudnet_amd", lib))
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_debug_build_audits_index_lists():
    assert os.path.exists(os.path.join(ROOT, "laudnet_amd", "libldn_hip_debug.so")), "build it: python -m laudnet_amd.build --debug"
    dbg = _run("libldn_hip_debug.so")
    rel = _run("libldn_hip.so")
    assert rel["clean"][0] == -1, "the release build must say that its checks are compiled away"
    assert dbg["initial"][0] == 0 and dbg["clean"] == [0, 0], f"index-bounds violations in a clean run: {dbg['clean']}"
    assert dbg["logits"] == rel["logits"], "debug and release builds must compute identical results"
    n, code = dbg["corrupted"]
    assert n > 0 and 300 <= code < 400, f"the corrupted channel list was not flagged by the fused tail's checks: {dbg['corrupted']}"


STALL_SCRIPT = r"""
import ctypes, json, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests/golden")
import torch
from laudnet_amd import _lib, ops
from fill import seeded_bernoulli
lib = _lib.load()
ops.set_math_mode("bf16x3")
stall = lib.ldn_debug_plan_stall
stall.argtypes, stall.restype = [ctypes.c_int], ctypes.c_int
B, S, Ho = 12, 7, 28
patch = seeded_bernoulli((B, S, S), 0.5, 7).cuda()
res = {"t0": ops.plan_timeouts(reset=True)}
good = ops.mask_to_index(patch, Ho, Ho, 1, patch_major=True)
torch.cuda.synchronize()
res["good_cnt"] = good.cnt.tolist()
res["t1"] = ops.plan_timeouts()
assert stall(5) == 0                      # image 5 never publishes: images 6 .. 11 run into the time bound
bad = ops.mask_to_index(patch, Ho, Ho, 1, patch_major=True)
torch.cuda.synchronize()
res["bad_cnt"] = bad.cnt.tolist()
res["bad_pre3"] = bad.pre3.tolist()
res["bad_pre1"] = bad.pre1.tolist()
res["bad_stats"] = bad.stats.tolist()
res["t2"] = ops.plan_timeouts()
try:
    ops.plan_timeouts(raise_on_error=True)
    res["raised"] = False
except _lib.LdnError:
    res["raised"] = True
# a consumer of the poisoned lists does no work and touches nothing -- and the failure is LOUD: the library's fault word is set, so the
# next library call's check() raises (VERDICT round 5, item 8) without anybody having polled the counter
h = torch.full((bad.cap3, 8), 3.0).cuda()
src = torch.randn(B * Ho * Ho, 8).cuda()
try:
    ops.conv_rows(src, torch.randn(8, 1, 8).cuda(), None, torch.zeros(8).cuda(), h, a_rows=bad.idx1, taps=1, m_count=bad.cnt[1:2], m_cap=bad.cap1)
    res["loud"] = False
except _lib.LdnError as e:
    res["loud"] = "bounded wait" in str(e)
torch.cuda.synchronize()
res["consumer_untouched"] = bool((h == 3.0).all())
assert stall(-1) == 0
res["t3"] = ops.plan_timeouts(reset=True)
again = ops.mask_to_index(patch, Ho, Ho, 1, patch_major=True)
torch.cuda.synchronize()
res["again_equal"] = bool(torch.equal(again.cnt, good.cnt) and torch.equal(again.pre3, good.pre3) and torch.equal(again.idx3[: int(good.cnt[0])], good.idx3[: int(good.cnt[0])]))
res["t4"] = ops.plan_timeouts()
print("RESULT " + json.dumps(res))
"""


def test_plan_timeout_leaves_empty_lists_and_is_counted():
    """a prefix wait of the one-launch list build (k_plan) that runs into its time bound must leave EMPTY lists -- counts,
    every prefix and the statistics zero, whatever the interleaving -- never uninitialised ones, and must be visible to the caller
    (ldn_plan_timeouts).  The debug build's hook makes one image never publish its counts; the bound is shortened to 150 ms."""
    env = dict(os.environ, LDN_LIB_PATH=os.path.join(ROOT, "laudnet_amd", "libldn_hip_debug.so"), LDN_PLAN_TIMEOUT_MS="150")
    p = subprocess.run([sys.executable, "-c", STALL_SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["t0"] == 0 and r["t1"] == 0 and r["good_cnt"][0] > 0
    assert r["bad_cnt"] == [0, 0], r["bad_cnt"]
    assert all(v == 0 for v in r["bad_pre3"]) and all(v == 0 for v in r["bad_pre1"]), "a prefix survived the failure"
    assert all(v == 0.0 for v in r["bad_stats"][:3])
    assert r["t2"] >= 1 and r["raised"] and r["t3"] == r["t2"]
    assert r["consumer_untouched"]
    assert r["loud"], "the call behind a failed list build must raise (ldn_fault_flag read by _lib.check)"
    assert r["again_equal"] and r["t4"] == 0, "the next launch must be healthy again"


CHAIN_STALL_SCRIPT = r"""
import ctypes, json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests", "golden"))
import torch
import laudnet_amd
import bench
from fill import fill_state_dict, seeded_randn
from laudnet_amd import _lib, ops
lib = _lib.load()
lib.ldn_debug_chain_stall.argtypes, lib.ldn_debug_chain_stall.restype = [ctypes.c_int], ctypes.c_int
ops.set_math_mode("bf16x3")
kw = dict(dyn_mode=["channel"] * 4, channel_dyn_granularity=[2, 2, 2, 2], channel_masker=["MLP"] * 4, channel_masker_layers=[2, 2, 2, 2],
          reduction_ratio=[16] * 4, num_classes=1000, input_size=224, width_mult=0.5)
m = laudnet_amd.uni_resnet50(**kw).eval()
sd = fill_state_dict(m.state_dict(), 3)
for k in sd:
    if k.endswith("bn3.weight"):
        sd[k] = sd[k] * 0.3
m.load_state_dict(sd)
m = m.cuda()
x = seeded_randn((9, 3, 224, 224), 77).cuda().contiguous(memory_format=torch.channels_last)
bench.calibrate_maskers(m, x, 0.62, None)
calls = []
orig = ops.bottleneck_chain
ops.bottleneck_chain = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
def fwd():
    with torch.no_grad():
        o = m(x, 1.0)
    torch.cuda.synchronize()
    return o[0].clone()
res = {"t0": ops.plan_timeouts(reset=True)}
good = fwd()
res["chained"] = len(calls)
res["t1"] = ops.plan_timeouts()
assert lib.ldn_debug_chain_stall(3) == 0          # image 3's loader never publishes: its seven consumer waves run into the bound once each
bad = fwd()                                       # (the launch terminates; the image's values are whatever had landed)
res["t2"] = ops.plan_timeouts()
res["others_equal"] = bool(torch.equal(bad[[0, 1, 2, 4, 5, 6, 7, 8]], good[[0, 1, 2, 4, 5, 6, 7, 8]]))
try:
    fwd()
    res["loud"] = False
except _lib.LdnError as e:
    res["loud"] = "bounded wait" in str(e)
torch.cuda.synchronize()
assert lib.ldn_debug_chain_stall(-1) == 0
res["t3"] = ops.plan_timeouts(reset=True)
again = fwd()
res["again_equal"] = bool(torch.equal(again, good))
res["t4"] = ops.plan_timeouts()
print("RESULT " + json.dumps(res))
"""


def test_chain_handoff_timeout_terminates_is_counted_and_loud():
    """The loader / consumer hand-off of k_chain_ld (csrc/ldn_chain_ld.h) spins on LDS words; every spin is bounded.  The debug build's hook makes one
    image's loader never publish: the launch must TERMINATE, the event must be counted (ldn_plan_timeouts) and LOUD (ldn_fault_flag: the next
    library call raises), the other images must be untouched, and the next launch must be healthy again."""
    env = dict(os.environ, LDN_LIB_PATH=os.path.join(ROOT, "laudnet_amd", "libldn_hip_debug.so"))
    p = subprocess.run([sys.executable, "-c", CHAIN_STALL_SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["chained"] >= 1, "stage 3 must have run as a chain"
    assert r["t0"] == 0 and r["t1"] == 0
    assert r["t2"] >= 1, "the failed hand-off waits must be counted"
    assert r["others_equal"], "the images whose hand-offs worked must be bit-identical to the healthy run"
    assert r["loud"], "the call behind a failed hand-off must raise (ldn_fault_flag read by _lib.check)"
    assert r["t3"] == r["t2"] and r["again_equal"] and r["t4"] == 0, "the next launch must be healthy again"
