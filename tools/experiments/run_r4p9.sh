#!/bin/bash
mkdir -p gpurun_out/r4p9
timeout 900 python -m pytest tests/test_hip_plan.py -x -q -m gpu > gpurun_out/r4p9/plan.log 2>&1; tail -12 gpurun_out/r4p9/plan.log
for w in layer regnet; do
for v in 1 0; do
LDN_FUSED_SPATIAL_MASKER=$v timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief > gpurun_out/r4p9/${w}_fused_$v.json 2> gpurun_out/r4p9/${w}_fused_$v.err
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4p9/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        de=d.get("dense_emulation_gpu",{})
        print(f.split("/")[-1], round(d["ms_per_step"],3), round(d.get("realised_speedup_vs_dense_emulation") or 0,3), de.get("max_abs_logit_diff_vs_hip_same_masks"))
    except Exception as e:
        print(f, "ERR", e)
P
