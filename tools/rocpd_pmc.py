#!/usr/bin/env python3
"""Print PMC counters per kernel dispatch from a rocprofv3 rocpd sqlite db.
usage: tools/rocpd_pmc.py results.db [name-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection "
                      "order by dispatch_id").fetchall()
    per = defaultdict(dict)
    meta = {}
    for d, k, g, c, v, dur in rows:
        if sub and sub not in k:
            continue
        per[d][c] = per[d].get(c, 0) + v
        meta[d] = (k.split("(")[0][-60:], g, dur)
    names = sorted({c for d in per.values() for c in d})
    print("dispatch kernel grid " + " ".join(names))
    for d in sorted(per):
        k, g, dur = meta[d]
        print(d, k, g, " ".join(f"{per[d].get(c, 0):.4g}" for c in names))


if __name__ == "__main__":
    main()
