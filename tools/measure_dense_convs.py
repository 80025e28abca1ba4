#!/usr/bin/env python3
"""Dense fp32 convolutions of ResNet-101 timed on this GPU through PyTorch-ROCm/MIOpen (channels_last, batch 256): the
measurements the latency predictor's free knobs are calibrated to (tools/predict_speedup.py).  Writes gpurun_out/r02_dense_convs.json."""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B = 256
shapes = [(1024, 256, 14, 1, 1), (256, 256, 14, 3, 1), (256, 1024, 14, 1, 1), (256, 64, 56, 1, 1), (64, 64, 56, 3, 1),
          (512, 128, 28, 1, 1), (128, 128, 28, 3, 1), (2048, 512, 7, 1, 1), (512, 512, 7, 3, 1)]
out = []
for cin, cout, h, ks, stride in shapes:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, ks, ks, device=dev).contiguous(memory_format=torch.channels_last)
    for _ in range(5):
        F.conv2d(x, w, None, stride, ks // 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        F.conv2d(x, w, None, stride, ks // 2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append(dict(cin=cin, cout=cout, h=h, ks=ks, stride=stride, ms=ms))
    print(out[-1], f"{2 * B * cin * cout * h * h * ks * ks / stride ** 2 / ms / 1e9:.1f} TFLOP/s", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"device": torch.cuda.get_device_name(0), "batch": B, "dtype": "fp32 (PyTorch-ROCm / MIOpen, channels_last, autotune on)",
           "convs": out}, open(os.path.join(ROOT, "gpurun_out", "r02_dense_convs.json"), "w"), indent=1)
