#!/bin/bash
timeout 900 python -m pytest tests/test_hip_adavit.py tests/test_hip_plan.py tests/test_hip_blocks.py -x -q -m gpu 2>&1 | tail -2
for w in adavit spatial layer channel; do
for v in 1 0 1 0; do
LDN_DENSE_MODEL=$v LDN_ROWS_HINT=$v timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w model=$v', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3))"
done; done
