#!/bin/bash
O=gpurun_out/r4x; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_blocks.py tests/test_hip_fullsize.py tests/test_hip_chain.py tests/test_hip_tail.py tests/test_det_golden.py -m gpu -x -q 2>&1 | tail -15
run() { env "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*'.split('bench.py')[0], round(d['ms_per_step'],3))" | tee -a $O/ab.log; }
run LDN_FUSED_F32=0 timeout 300 python bench.py --math fp32 --no-legs --steps 5 --warmup 2 --keep 0.6066 2>/dev/null
run LDN_FUSED_F32=1 timeout 300 python bench.py --math fp32 --no-legs --steps 5 --warmup 2 --keep 0.6066 2>/dev/null
