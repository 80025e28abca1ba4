#!/bin/bash
# round 4, call d: four-wave workgroups for the 128-wide stride-1 tail -- parity, A/B
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_tail.py tests/test_hip_chain.py tests/test_hip_fullsize.py tests/test_hip_blocks.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
  for f in 0 1; do
    LDN_TAIL_NW4=$f timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nw4=$f', round(d['ms_per_step'],3))" | tee -a $O/ab.log
  done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 5 --warmup 2 --no-legs --keep 0.6066 > $R/$O/prof.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_c/*.db | head -1) 24 "naive_conv|igemm_|Cijk" > $R/$O/kernel_stats.txt 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $R/$O/period.txt 2>&1
grep -n "k_tail<4" $R/$O/period.txt
