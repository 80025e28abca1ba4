#!/usr/bin/env python3
"""One whole step (from a stem launch (k_stem, else the max-pool) to the next) of a rocprofv3 (rocpd sqlite) kernel trace: every kernel in start order;
runs of kernels shorter than min_us are folded into one line (count, busy time, span).  usage: rocpd_period.py results.db [min_us] [marker kernel] [its launches per step]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
marker = sys.argv[3] if len(sys.argv) > 3 else None          # (workloads without a stem: a kernel that runs `per` times per step, e.g. k_packed_mha 12)
per = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if marker:
    pools = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = pools[-1 - 2 * per], pools[-1 - per]
else:
    pools = [i for i, r in enumerate(rows) if "k_stem" in r[0]] or [i for i, r in enumerate(rows) if "max_pool" in r[0]]
    a, b = pools[-2], pools[-1]
rows = rows[a:b]
t0 = rows[0][1]
prev_end = t0
small = []
busy = 0.0


def flush():
    global small
    if small:
        s0, e1 = small[0][1], max(r[2] for r in small)
        names = {}
        for r in small:
            k = r[0].replace("void ", "").replace("at::native::", "").replace("ldn::", "")[:40]
            names[k] = names.get(k, 0) + 1
        top = ", ".join(f"{v}x {k}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4])
        print(f"{(s0 - t0) / 1e3:10.1f} us  [{len(small):3d} small kernels: busy {sum(r[2] - r[1] for r in small) / 1e3:7.1f} us over {(e1 - s0) / 1e3:7.1f} us]  {top}")
        small = []


for r in rows:
    name, s, e = r
    dur = (e - s) / 1e3
    busy += dur
    if dur < min_us:
        small.append(r)
    else:
        flush()
        short = name.replace("void ", "").replace("ldn::", "")[:70]
        print(f"{(s - t0) / 1e3:10.1f} us  +{dur:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short}")
    prev_end = max(prev_end, e)
flush()
print(f"period {(rows[-1][2] - t0) / 1e3:.1f} us, busy {busy:.1f} us, kernels {len(rows)}")
