#!/usr/bin/env python3
"""profiles/rNN_pmc_chain.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes over bench.py) -> profiles/rNN_traffic.json,
the file bench.py reads `roofline.traffic` from.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md).
usage: tools/make_traffic_json.py profiles/r03_pmc_chain.txt profiles/r03_traffic.json [date]"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
date = sys.argv[3] if len(sys.argv) > 3 else ""
sec, vals = None, {}
for line in open(src):
    if line.startswith("=="):
        sec = line[2:].split("(")[0].split()
        continue
    m = re.match(r"\d+ .*k_chain<\d+(?:, \w+)?> \d+ (.*)", line)
    if m and sec:
        nums = [float(v) for v in m.group(1).split()]
        for name, v in zip(sorted(sec), nums):
            vals.setdefault(name, []).append(v)
fetch = min(vals["FETCH_SIZE"])      # KiB per dispatch, steady-state launches (the first one also pulls the weights from HBM)
write = min(vals["WRITE_SIZE"])
out = {"chain_fused_bf16x3": {
    "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
    "fetch_kib_raw": fetch, "write_kib": write,
    "scope": "whole k_chain launch = 22 stage-3 blocks (9.14 GB algorithmic: every block's input read once + output written once); "
             "FETCH_SIZE/WRITE_SIZE are counted at the L2-fabric boundary, so Infinity-Cache hits are included",
    "source": f"{src} ({date}: separate rocprofv3 --kernel-trace --pmc passes over `bench.py --no-legs`; FETCH_SIZE doubled per "
              "MI355X_MICROARCH.md; committed file, NOT measured in the bench run that quotes it)"}}
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
    out["chain_fused_bf16x3"]["mfma_busy_cycles"] = min(vals["SQ_VALU_MFMA_BUSY_CYCLES"])
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
