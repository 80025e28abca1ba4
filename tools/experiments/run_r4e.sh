#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
timeout 600 python tools/experiments/pipe_check.py 2>&1 | grep -v amdgpu.ids | tee $O/check.log
for i in 1 2; do
  for n in 1 2 4; do
    LDN_BATCH_PIPE=$n timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipe=$n', round(d['ms_per_step'],3))" | tee -a $O/ab.log
  done
done
for w in spatial layer; do for n in 1 2; do
    LDN_BATCH_PIPE=$n timeout 300 python bench.py --workload $w --no-legs --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w pipe=$n', round(d['ms_per_step'],3))" | tee -a $O/ab.log
done; done
