#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_tail.py tests/test_hip_chain.py tests/test_hip_fullsize.py tests/test_hip_blocks.py tests/test_hip_ops.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', round(d['ms_per_step'],3))"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 5 --warmup 2 --no-legs --keep 0.6066 > $R/$O/prof.log 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $R/$O/period.txt 2>&1
grep -E "k_tail<4|k_head<4" $R/$O/period.txt
