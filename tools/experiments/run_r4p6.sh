#!/bin/bash
mkdir -p gpurun_out/r4p6
for i in 1 2; do
timeout 600 python bench.py --workload spatial --no-cpu --no-secondary --no-pmc > gpurun_out/r4p6/spatial_on$i.json 2> gpurun_out/r4p6/spatial_on$i.err
LDN_FUSED_SPATIAL_MASKER=0 timeout 600 python bench.py --workload spatial --no-cpu --no-secondary --no-pmc > gpurun_out/r4p6/spatial_planonly$i.json 2> gpurun_out/r4p6/spatial_planonly$i.err
LDN_FUSED_SPATIAL_MASKER=0 LDN_INDEX_PLAN=0 timeout 600 python bench.py --workload spatial --no-cpu --no-secondary --no-pmc > gpurun_out/r4p6/spatial_off$i.json 2> gpurun_out/r4p6/spatial_off$i.err
done
timeout 600 python bench.py --workload regnet --no-cpu --no-secondary --no-pmc > gpurun_out/r4p6/regnet_on.json 2> gpurun_out/r4p6/regnet_on.err
LDN_INDEX_PLAN=0 timeout 600 python bench.py --workload regnet --no-cpu --no-secondary --no-pmc > gpurun_out/r4p6/regnet_off.json 2> gpurun_out/r4p6/regnet_off.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4p6/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        de=d.get("dense_emulation_gpu",{})
        print(f.split("/")[-1], round(d["ms_per_step"],3), round(d.get("realised_speedup_vs_dense_emulation",0),3), de.get("max_abs_logit_diff_vs_hip_same_masks"), d["config"].get("launch"), d.get("graph_replay",{}) if isinstance(d.get("graph_replay"),dict) else None)
    except Exception as e:
        print(f, "ERR", e)
P
