#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total/avg/min/max duration, share.
usage: tools/rocpd_stats.py results.db [top_n] [exclude_regex]
exclude_regex drops kernels by name before the shares are computed (e.g. MIOpen's find-mode trial kernels:
'naive_conv|igemm_|grouped_conv_fwd|SubTensorOp|Im2d2Col|gemm|Cijk')."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    if len(sys.argv) > 3:
        rx = re.compile(sys.argv[3])
        rows = [r for r in rows if not rx.search(r[0])]
    total = sum(r[2] for r in rows)
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for name, n, tot, avg, mn, mx in rows[:top]:
        print(f"{name[:90]:90s} {n:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}")
    print(f"{'TOTAL':90s} {sum(r[1] for r in rows):7d} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main()
