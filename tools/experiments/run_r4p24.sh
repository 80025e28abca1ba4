#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu -k "chain or tail or head or block or model or small or folded" 2>&1 | tail -2
for v in pre nopre pre nopre pre nopre; do
if [ $v = nopre ]; then export LDN_LIB_PATH=$PWD/tools/ablate/libldn_nopreh.so; else unset LDN_LIB_PATH; fi
timeout 600 python bench.py --workload channel --steps 20 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('channel $v', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3))"
done
