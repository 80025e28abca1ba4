#!/bin/bash
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py -m gpu -x -q 2>&1 | tail -2
run() { w=$1; shift; env "$@" timeout 300 python bench.py --workload $w --no-legs --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $*', round(d['ms_per_step'],3))"; }
for i in 1 2; do for w in layer spatial regnet channel; do run $w A=new; run $w LDN_LIB_PATH=tools/ablate/libldn_base.so; done; done
