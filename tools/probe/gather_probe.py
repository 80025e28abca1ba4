"""tuning only: per-CU LDS-DMA rate from an L2-resident region read by every workgroup, by gather granularity (tools/probe/gather_probe.hip)."""
import ctypes, os
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "ablate", "libgather_probe.so"))
lib.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
names = {0: "1 KB contiguous per instruction", 1: "16-B pieces, every other one", 2: "16-B pieces, random slot of 4", 3: "32-B runs, random slot of 4", 4: "64-B runs, random slot of 4"}
print("region   pattern                              useful GB/s per CU   TB/s chip")
for region_mb in (2, 16):
    region = region_mb << 20
    buf = torch.zeros(region, dtype=torch.uint8, device=dev)
    for pattern in (0, 1, 2, 3, 4):
        row = {0: 1024, 1: 2048, 2: 4096, 3: 8192, 4: 16384}[pattern]
        reps = 16 if region_mb == 2 else 2
        go = lambda: lib.gather_probe(buf.data_ptr(), region, 256, reps, pattern, sink.data_ptr(), st)
        assert go() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        useful = (region // row) * 1024 * reps        # bytes one workgroup moved
        per_cu = useful / (ms * 1e-3) / 1e9
        print(f"{region_mb:3d} MB   {names[pattern]:36s} {per_cu:10.1f}          {per_cu * 256 / 1e3:6.2f}")
