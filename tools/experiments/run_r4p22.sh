#!/bin/bash
timeout 900 python -m pytest tests/test_hip_adavit.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --workload adavit --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adavit', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3), d.get('dense_emulation_gpu',{}).get('max_abs_diff_vs_hip_same_masks'))"
done
