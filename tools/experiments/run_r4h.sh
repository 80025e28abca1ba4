#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_stem.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 5 --warmup 2 --no-legs --keep 0.6066 > $R/$O/prof.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_c/*.db | head -1) 24 "naive_conv|igemm_|Cijk" > $R/$O/kernel_stats.txt 2>&1
grep -E "k_stem|k_chain|k_tail<2, 1, false|k_smallmap" $R/$O/kernel_stats.txt
cd $R; for i in 1 2; do timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', round(d['ms_per_step'],3))"; done
