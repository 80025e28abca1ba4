#!/bin/bash
# Step time of the bench workloads as a function of the maskers' keep probability: the calibration / validation data of
# laudnet_amd/predictor.py (run on the GPU box: gpurun -- 'bash tools/density_sweep.sh [workloads]'; ROUND_TAG names the files, default r06).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sweep
mkdir -p $OUT
for w in ${@:-channel spatial layer regnet}; do
  rm -f $OUT/${ROUND_TAG:-r06}_density_sweep_$w.jsonl
  for k in 0.25 0.4 0.5 0.62 0.75 0.9 1.0; do
    python $R/bench.py --workload $w --no-legs --steps 8 --warmup 2 --keep $k 2>/dev/null | tail -1 >> $OUT/${ROUND_TAG:-r06}_density_sweep_$w.jsonl
  done
  python - <<PY
import json
for l in open("$OUT/${ROUND_TAG:-r06}_density_sweep_$w.jsonl"):
    d = json.loads(l)
    r = d.get("roofline") or {}
    print("$w", d["config"]["workload"][-12:], "ms/step %.2f" % d["ms_per_step"], "flops ratio", d["config"]["mean_block_flops_ratio"],
          "chain us/block", r.get("avg_us_per_block"))
PY
done
