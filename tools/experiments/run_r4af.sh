#!/bin/bash
mkdir -p gpurun_out/r4af
t0=$(date +%s)
timeout 1500 python bench.py > gpurun_out/r4af/bench_default.json 2> gpurun_out/r4af/bench_default.err
echo "wall $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4af/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(round(d["ms_per_step"],3), round(d["realised_speedup_vs_dense_emulation"],3), r["traffic"], r["traffic_source"][:70], round(r["frac"],3))
print(d["cpu_baseline"]["cores"], round(d["cpu_baseline"]["value"],1), list(d["secondary"].keys()))
PY
tail -c 400 gpurun_out/r4af/bench_default.err
