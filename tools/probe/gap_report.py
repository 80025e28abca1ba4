"""gaps behind every kernel of tools/probe/gap_probe (rocpd sqlite): python gap_report.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
for i in range(len(rows) - 1):
    n, s, e = rows[i]
    if "tiny" in n and "tiny" not in rows[i - 1][0]:
        continue
    print(f"{n[:40]:40s} dur {(e - s) / 1e3:9.1f} us   gap behind it {(rows[i + 1][1] - e) / 1e3:6.1f} us")
