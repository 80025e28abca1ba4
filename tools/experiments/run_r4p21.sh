#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu -k "rows or dense or block or model or adavit or regnet or plan or small or tail" 2>&1 | tail -2
for w in spatial adavit layer channel regnet; do
for v in pre nopre pre nopre; do
if [ $v = nopre ]; then export LDN_LIB_PATH=$PWD/tools/ablate/libldn_nopre.so; else unset LDN_LIB_PATH; fi
timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $v', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3))"
done; done
