#!/usr/bin/env python3
"""Tuning: HOST-side duration of every C-ABI call during one forward of a bench workload (does a launch block until the previous kernel has
finished?).  usage: python tools/host_timeline.py [workload] [--no-devkernarg]"""
import os
import sys
import time

if "--no-devkernarg" in sys.argv:
    os.environ["HIP_FORCE_DEV_KERNARG"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
import bench  # noqa: E402
import laudnet_amd  # noqa: E402
from laudnet_amd import _lib, ops  # noqa: E402
from fill import fill_state_dict, seeded_randn  # noqa: E402

wname = next((a for a in sys.argv[1:] if not a.startswith("--")), "spatial")
wl = bench.WORKLOADS[wname]
dev = torch.device("cuda:0")
ops.set_math_mode("bf16x3")
kw = dict(wl["kw"], num_classes=1000, input_size=224)
model = getattr(laudnet_amd, wl.get("arch", "uni_resnet101"))(**kw).eval()
sd = fill_state_dict(model.state_dict(), 1)
for k in sd:
    if k.endswith("bn3.weight"):
        sd[k] = sd[k] * 0.3
model.load_state_dict(sd)
model = model.to(dev)
x = seeded_randn((256, 3, 224, 224), 1000).to(dev).contiguous(memory_format=torch.channels_last)
bench.calibrate_maskers(model, x, wl["p_channel"], wl["p_spatial"])
with torch.no_grad():
    for _ in range(4):
        model(x, 1.0)
torch.cuda.synchronize()
lib = _lib.load()
log = []
for name in _lib.SIGNATURES:
    fn = getattr(lib, name)

    def wrapped(*a, _fn=fn, _name=name):
        t0 = time.perf_counter()
        r = _fn(*a)
        log.append((_name, t0, time.perf_counter()))
        return r
    setattr(lib, name, wrapped)
with torch.no_grad():
    t_begin = time.perf_counter()
    model(x, 1.0)
    t_host = time.perf_counter()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
print(f"{wname}: host finished issuing after {1e3 * (t_host - t_begin):.2f} ms, GPU finished after {1e3 * (t_end - t_begin):.2f} ms; {len(log)} library calls")
import collections
agg = collections.defaultdict(list)
for n, a, b in log:
    agg[n].append(1e6 * (b - a))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"  {n:32s} calls {len(v):4d}  host us per call: median {v2[len(v2) // 2]:8.1f}  max {v2[-1]:8.1f}  total {sum(v) / 1e3:7.2f} ms")
prev = t_begin
print("first 40 calls: name, host gap before the call (us), call duration (us)")
for n, a, b in log[:40]:
    print(f"  {n:32s} {1e6 * (a - prev):8.1f} {1e6 * (b - a):8.1f}")
    prev = b
