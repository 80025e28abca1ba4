#!/usr/bin/env python3
"""Tuning experiment: the headline forward as two half-batches on two HIP streams (22.5 ms) vs one full batch (23.2 ms) vs the
two halves back to back (30.8 ms) -- concurrency between independent launches buys ~3 %, not enough to restructure for."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import laudnet_amd
from laudnet_amd import ops
from fill import fill_state_dict, seeded_randn
import bench
dev = torch.device('cuda:0')
ops.set_math_mode('bf16x3')
wl = bench.WORKLOADS['channel']
kw = dict(wl['kw'], num_classes=1000, input_size=224)
def make():
    m = laudnet_amd.uni_resnet101(**kw).eval()
    sd = fill_state_dict(m.state_dict(), 1)
    for k in sd:
        if k.endswith('bn3.weight'): sd[k] = sd[k] * 0.3
    m.load_state_dict(sd); return m.to(dev)
m1 = make()
x = seeded_randn((256, 3, 224, 224), 1000).to(dev).contiguous(memory_format=torch.channels_last)
bench.calibrate_maskers(m1, x, wl['p_channel'], wl['p_spatial'])
m2 = make(); m2.load_state_dict(m1.state_dict())
def t(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print('full batch 256      :', t(lambda: m1(x, 1.0)))
    xa, xb = x[:128].contiguous(memory_format=torch.channels_last), x[128:].contiguous(memory_format=torch.channels_last)
    print('two halves, 1 stream:', t(lambda: (m1(xa, 1.0), m2(xb, 1.0))))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): m1(xa, 1.0)
        with torch.cuda.stream(s2): m2(xb, 1.0)
        cur.wait_stream(s1); cur.wait_stream(s2)
    print('two halves, 2 streams:', t(both))
