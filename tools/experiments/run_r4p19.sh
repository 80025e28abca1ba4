#!/bin/bash
for c in "1700,500,12000" "1700,500,20000" "1000,500,12000" "1700,500,6000" "1700,500,12000"; do
for w in spatial adavit layer; do
LDN_DENSE_MODEL_C=$c timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $w', round(d['ms_per_step'],3))"
done; done
