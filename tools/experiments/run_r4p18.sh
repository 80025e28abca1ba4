#!/bin/bash
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for w in regnet spatial layer adavit; do
for v in 1 0 1 0; do
LDN_DENSE_MODEL=$v LDN_ROWS_HINT=$v timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w model=$v', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3))"
done; done
