#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --stats -d /tmp/pg -o r -- python $R/bench.py --workload regnet --steps 3 --warmup 2 --no-legs > /dev/null 2>&1
python - $(ls /tmp/pg/*.db | head -1) <<'P'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
# last forward only: from the last k_stem3 on
idx = [i for i, r in enumerate(rows) if "k_stem3" in r[0]]
a, b = idx[-2], idx[-1]
gaps = collections.defaultdict(list)
for i in range(a, b - 1):
    n = rows[i][0].replace("void ", "").replace("ldn::", "")[:48]
    gaps[n].append((rows[i + 1][1] - rows[i][2]) / 1e3)
print("gap BEHIND each kernel (us): name, launches, mean gap, min, max")
for n, g in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:50s} {len(g):4d} {sum(g)/len(g):7.2f} {min(g):7.2f} {max(g):7.2f}")
print("period us", (rows[b][1] - rows[a][1]) / 1e3, "busy", sum((r[2] - r[1]) for r in rows[a:b]) / 1e3)
P
