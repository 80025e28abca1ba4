#!/usr/bin/env python3
"""What does this box sustain for streaming traffic of the size of a stage-1 tensor (822 MB fp32)?  torch copy (read + write), read-only reduction, fill (write only), add (2 reads + 1 write)."""
import json
import time

import torch

dev = torch.device("cuda", 0)
n = 256 * 56 * 56 * 256
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
z = torch.empty_like(x)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {}
b = 4 * n
res["copy_TBps"] = 2 * b / t(lambda: y.copy_(x)) / 1e12
res["read_sum_TBps"] = b / t(lambda: x.sum()) / 1e12
res["fill_TBps"] = b / t(lambda: y.fill_(1.0)) / 1e12
res["add_2r1w_TBps"] = 3 * b / t(lambda: torch.add(x, y, out=z)) / 1e12
res["relu_add_inplace_1r_1rw_TBps"] = 3 * b / t(lambda: y.add_(x)) / 1e12
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
