#!/bin/bash
mkdir -p gpurun_out/r4p3
timeout 900 python -m pytest tests/test_hip_plan.py -x -q -m gpu > gpurun_out/r4p3/plan.log 2>&1; tail -3 gpurun_out/r4p3/plan.log
for i in 1 2; do
timeout 600 python bench.py --workload spatial --no-cpu --no-secondary --no-pmc > gpurun_out/r4p3/spatial_on$i.json 2> gpurun_out/r4p3/spatial_on$i.err
LDN_FUSED_SPATIAL_MASKER=0 timeout 600 python bench.py --workload spatial --no-cpu --no-secondary --no-pmc > gpurun_out/r4p3/spatial_planonly$i.json 2> gpurun_out/r4p3/spatial_planonly$i.err
LDN_FUSED_SPATIAL_MASKER=0 LDN_INDEX_PLAN=0 timeout 600 python bench.py --workload spatial --no-cpu --no-secondary --no-pmc > gpurun_out/r4p3/spatial_off$i.json 2> gpurun_out/r4p3/spatial_off$i.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4p3/spatial_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        de=d.get("dense_emulation_gpu",{})
        print(f.split("/")[-1], round(d["ms_per_step"],3), round(d.get("realised_speedup_vs_dense_emulation",0),3), de.get("max_abs_logit_diff_vs_hip_same_masks"), d["config"].get("launch"))
    except Exception as e:
        print(f, "ERR", e)
P
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4p3/full.log 2>&1; tail -3 gpurun_out/r4p3/full.log
