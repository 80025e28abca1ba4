#!/usr/bin/env python3
"""One forward pass as a timeline from a rocprofv3 (rocpd sqlite) kernel trace: every kernel after the LAST max-pool launch,
in start order, with start offset, duration and the gap to the previous kernel's end (tuning; usage: rocpd_timeline.py results.db [min_us])."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in ("grid_x", "grid_size_x", "workgroup_x", "stream_id", "queue_id") if c in cols]
rows = db.execute(f"select {name_col}, start, end {''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
last = max(i for i, r in enumerate(rows) if "max_pool" in r[0])
rows = rows[last:]
t0 = rows[0][1]
prev_end = t0
tot = 0.0
for r in rows:
    name, s, e = r[0], r[1], r[2]
    dur = (e - s) / 1e3
    tot += dur
    if dur >= min_us:
        short = name.replace("void ", "").replace("ldn::", "")[:60]
        print(f"{(s - t0) / 1e3:10.1f} us  +{dur:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short:60s} {' '.join(str(x) for x in r[3:])}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {tot:.1f} us, kernels {len(rows)}")
