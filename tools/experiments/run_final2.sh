#!/bin/bash
mkdir -p gpurun_out/final_r4
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/final_r4/full_gpu_tests.log 2>&1; tail -2 gpurun_out/final_r4/full_gpu_tests.log
timeout 1500 python bench.py > gpurun_out/final_r4/bench_default.json 2> gpurun_out/final_r4/bench_default.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/final_r4/bench_default.json").read().strip().splitlines()[-1])
print("headline", round(d["ms_per_step"],3), d.get("realised_speedup_vs_dense_emulation"), d["roofline"]["frac"], d["roofline"].get("traffic"), d["cpu_baseline"]["value"])
for k,v in d.get("secondary",{}).items():
    print(k, v.get("ms_per_step"), v.get("realised_speedup_vs_dense_emulation"), v.get("max_abs_diff_vs_oracle_same_masks"), v.get("error"))
P
