#!/bin/bash
for w in layer spatial; do
for v in 1 0 1 0; do
LDN_FUSED_SPATIAL_MASKER=$v timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w fused=$v', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3))"
done; done
