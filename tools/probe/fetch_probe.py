"""tuning only: per-CU fetch rate of one 512-thread workgroup per CU against the bytes it keeps in flight (tools/probe/fetch_probe.hip)."""
import ctypes, os, sys
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "ablate", "libfetch_probe.so"))
lib.fetch_probe.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(wgs, bytes_per_wg, reps, depth, mode, shared):
    n = bytes_per_wg if shared else wgs * bytes_per_wg
    buf = torch.zeros(n, dtype=torch.uint8, device=dev)
    def go():
        # the kernel offsets by blockIdx * bytes_per_wg: for the shared case launch with a region all workgroups alias through a zero stride
        rc = lib.fetch_probe(buf.data_ptr(), bytes_per_wg, wgs, reps, depth, mode, 0, sink.data_ptr(), st) if not shared else \
            lib.fetch_probe(buf.data_ptr(), 0, wgs, reps, depth, mode, 0, sink.data_ptr(), st)
        assert rc == 0, rc
    return go


if __name__ == "__main__":
    print("wgs  region/WG  in-flight/CU  mode      GB/s per CU   TB/s chip   (implied latency us = in-flight / rate)")
    for wgs in (256, 128, 32):
        for bytes_per_wg, reps in ((800 * 1024, 8),):
            for mode, depth in ((0, 1), (0, 2), (0, 4), (0, 8), (0, 12), (0, 16), (1, 1)):
                go = timed(wgs, bytes_per_wg, reps, depth, mode, False)
                go(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); go(); e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                per_cu = bytes_per_wg * reps / (ms * 1e-3) / 1e9
                infl = 8 * depth
                print(f"{wgs:4d} {bytes_per_wg // 1024:6d} KB {infl:8d} KB   {'lds-dma' if mode == 0 else 'vgpr   '}  {per_cu:10.1f}   {per_cu * wgs / 1e3:8.2f}     {infl * 1024 / (per_cu * 1e9) * 1e6 if mode == 0 else float('nan'):6.2f}")
