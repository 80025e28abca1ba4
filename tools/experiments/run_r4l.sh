#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))" | tee -a $O/ab.log; }
for i in 1 2; do
  run A=0
  run LDN_DENSE_CHANNEL_3X3=1
  run LDN_SIDE_STREAM=1
  run LDN_DENSE_CHANNEL_3X3=1 LDN_SIDE_STREAM=1
done
