"""Does HIP_FORCE_DEV_KERNARG take effect when set after `import torch` (before the first HIP call)?  python kernarg_probe.py early|late|off"""
import os, sys, time
mode = sys.argv[1]
if mode == "early":
    os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
elif mode == "off":
    os.environ["HIP_FORCE_DEV_KERNARG"] = "0"
import torch
if mode == "late":
    os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    e0.record()
    for _ in range(2000):
        x.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
print(mode, "us per tiny kernel:", round(e0.elapsed_time(e1) / 2000 * 1e3, 2))
