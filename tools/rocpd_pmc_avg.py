#!/usr/bin/env python3
"""PMC counters of a rocprofv3 rocpd sqlite db averaged per (kernel, grid): one line per launch shape.
usage: tools/rocpd_pmc_avg.py results.db [name-substring[,name-substring...]]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    subs = [s for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if s]
    rows = db.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection "
                      "order by dispatch_id").fetchall()
    per = defaultdict(dict)
    meta = {}
    for d, k, g, c, v, dur in rows:
        if subs and not any(s in k for s in subs):
            continue
        per[d][c] = per[d].get(c, 0) + v
        meta[d] = (k.split("(")[0].replace("void ", "").replace("ldn::", "")[-48:], g, dur)
    names = sorted({c for d in per.values() for c in d})
    groups = defaultdict(list)
    for d in per:
        groups[meta[d][:2]].append(d)
    print("kernel grid launches avg_dur_us " + " ".join(names))
    for (k, g), ds in sorted(groups.items()):
        n = len(ds)
        dur = sum(meta[d][2] for d in ds) / n / 1e3
        print(k, g, n, f"{dur:.1f}", " ".join(f"{sum(per[d].get(c, 0) for d in ds) / n:.4g}" for c in names))


if __name__ == "__main__":
    main()
