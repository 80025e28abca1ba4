#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { env "$@" timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))" | tee -a $O/ab.log; }
for i in 1 2 3; do
  run A=0
  run LDN_DENSE_CHANNEL_3X3=0 LDN_SIDE_STREAM=0
done
run LDN_FOLD_PROJ=0 LDN_STEM_GAP=0 LDN_DENSE_CHANNEL_3X3=0 LDN_SIDE_STREAM=0
