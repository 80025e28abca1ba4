"""tuning only: what plain streaming kernels reach on this GPU at the tensor sizes of stages 1-2 (the ceilings the HBM-bound launches are compared with)."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, shape in (("stage-1 map [256,56,56,256]", (256, 56, 56, 256)), ("stage-2 map [256,28,28,512]", (256, 28, 28, 512)), ("stage-3 map [256,14,14,1024]", (256, 14, 14, 1024))):
    x = torch.randn(shape, device=dev); y = torch.empty_like(x); z = torch.randn(shape, device=dev)
    mb = x.numel() * 4 / 1e6
    us = t(lambda: torch.relu_(x));                 print(f"{name}: in-place relu   (read + write {2*mb:7.0f} MB) {us:7.1f} us  {2*mb/us*1e-6*1e6/1e3:6.2f} TB/s")
    us = t(lambda: y.copy_(x));                     print(f"{name}: copy            (read + write {2*mb:7.0f} MB) {us:7.1f} us  {2*mb/us/1e3:6.2f} TB/s")
    us = t(lambda: torch.add(x, z, out=y));         print(f"{name}: y = x + z       (2 reads + write {3*mb:7.0f} MB) {us:7.1f} us  {3*mb/us/1e3:6.2f} TB/s")
    us = t(lambda: x.sum());                        print(f"{name}: sum             (read {mb:7.0f} MB) {us:7.1f} us  {mb/us/1e3:6.2f} TB/s")
