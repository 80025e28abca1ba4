#!/bin/bash
mkdir -p gpurun_out/r4p8
timeout 1500 python -m pytest tests -x -q -m gpu -k "rows or dense or block or model or tail or small or adavit or regnet or plan" > gpurun_out/r4p8/tests.log 2>&1; tail -3 gpurun_out/r4p8/tests.log
for w in spatial layer channel; do
for v in 1 0; do
LDN_DENSE_R128=$v timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief > gpurun_out/r4p8/${w}_r128_$v.json 2> gpurun_out/r4p8/${w}_r128_$v.err
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4p8/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        de=d.get("dense_emulation_gpu",{})
        print(f.split("/")[-1], round(d["ms_per_step"],3), round(d.get("realised_speedup_vs_dense_emulation") or 0,3), de.get("max_abs_logit_diff_vs_hip_same_masks"))
    except Exception as e:
        print(f, "ERR", e)
P
