#!/bin/bash
timeout 900 python -m pytest tests/test_hip_tail.py tests/test_hip_chain.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -2
LDN_LIB_PATH=tools/ablate/libldn_trace.so timeout 300 python tools/trace_head.py 3 2>&1 | grep -v amdgpu.ids
run() { env "$@" timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run A=wlate; run LDN_LIB_PATH=tools/ablate/libldn_wburst.so; done
