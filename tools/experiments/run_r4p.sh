#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py tests/test_hip_fullsize.py tests/test_det_golden.py -m gpu -x -q 2>&1 | tail -4
run() { env "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*'.split('bench.py')[-1], round(d['ms_per_step'],3))" | tee -a $O/ab.log; }
for w in channel spatial layer regnet; do
  run LDN_DENSE_F32=0 timeout 300 python bench.py --workload $w --math fp32 --no-legs --steps 5 --warmup 2 2>/dev/null
  run LDN_DENSE_F32=1 timeout 300 python bench.py --workload $w --math fp32 --no-legs --steps 5 --warmup 2 2>/dev/null
done
