#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_tail.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --math fp32 --steps 3 --warmup 1 --no-legs --keep 0.6066 > $R/$O/prof.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_c/*.db | head -1) 16 "naive_conv|igemm_|Cijk" > $R/$O/kernel_stats_fp32.txt 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 40 > $R/$O/period_fp32.txt 2>&1
head -18 $R/$O/kernel_stats_fp32.txt
