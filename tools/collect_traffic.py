#!/usr/bin/env python3
"""On the GPU box: PMC traffic of the dominant kernel kinds of every workload -> one JSON (the file bench.py quotes as profiles/rNN_traffic.json).
usage: tools/collect_traffic.py <out.json>      (runs bench.py --brief --pmc-legs per secondary workload, and the k_chain / k_tail passes of the headline)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out = {}
date = time.strftime("%Y-%m-%d")
for w in ("spatial", "layer", "regnet"):
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--brief", "--pmc-legs", "--no-cpu"], capture_output=True, text=True, timeout=1200)
    try:
        d = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
        r = d["roofline"]
        kind = None
        for k, pats in bench.KIND_KERNELS.items():
            if r.get("traffic_detail", {}).get("kernels") == list(pats):
                kind = k
        out[f"{w}:{kind}"] = {"traffic_bytes_per_launch": r["traffic"], "scope": r.get("traffic_scope"),
                              "source": f"{date}, tools/collect_traffic.py: " + (r.get("traffic_source") or "").replace("MEASURED IN THIS RUN: ", ""),
                              "algorithmic_mbytes_per_launch": r.get("algorithmic_mbytes_per_launch"), "traffic_over_algorithmic": r.get("traffic_over_algorithmic"),
                              **(r.get("traffic_detail") or {})}
        keep = d["config"].get("keep_probability_calibrated_to")
        print(w, kind, out[f"{w}:{kind}"]["traffic_bytes_per_launch"], r.get("traffic_over_algorithmic"), "keep", keep, flush=True)
    except Exception as e:
        print(w, "failed:", repr(e)[:200], pr.stderr[-500:], flush=True)
# headline: the chained launch and the fused tails (keep from a brief run's calibration)
pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--brief", "--no-cpu", "--no-pmc"], capture_output=True, text=True, timeout=1200)
d = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
keep = d["config"]["keep_probability_calibrated_to"]
ch = bench.measure_chain_traffic(keep, 256)
if ch:
    ch["source"] = f"{date}, tools/collect_traffic.py: " + ch["source"].replace("MEASURED IN THIS RUN on this box: ", "")
    out["chain_fused_bf16x3"] = ch
tl = bench.measure_kind_traffic("channel", "tail_fused", keep, 256)
if tl:
    tl["source"] = f"{date}, tools/collect_traffic.py: " + tl["source"]
    out["tail_fused_bf16x3"] = tl
print("chain", ch and ch["traffic_bytes_per_launch"], "tail", tl and tl["traffic_bytes_per_launch"], flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
