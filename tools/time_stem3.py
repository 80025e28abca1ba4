"""Times ldn_stem3_conv (LAD-RegNet stem) at bs256 / 224^2 against the library conv -> BN -> ReLU (tuning aid)."""
import torch, torch.nn as nn
from laudnet_amd import ops, load_library
load_library()
ops.set_math_mode("bf16x3")
dev = "cuda:0"
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
conv = nn.Conv2d(3, 32, 3, 2, 1, bias=False).to(dev)
bn = nn.BatchNorm2d(32).to(dev).eval()
frag = ops.pack_stem3_weights(conv.weight.detach())
shift = torch.zeros(32, device=dev)
xn = ops.as_nhwc(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
with torch.no_grad():
    print("k_stem3 us", t(lambda: ops.stem3_conv(xn, frag, shift, 32)))
    print("library conv+bn+relu us", t(lambda: torch.relu_(bn(conv(x)))))
    print("library conv us", t(lambda: conv(x)))
