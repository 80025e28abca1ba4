#!/bin/bash
# round 4, call c: folded projection with the tiled x_split + stem GAP -- parity, A/B, per-kernel times
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_tail.py tests/test_hip_stem.py tests/test_hip_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
  for cfg in "0 0" "1 0" "1 1"; do
    set -- $cfg
    LDN_FOLD_PROJ=$1 LDN_STEM_GAP=$2 timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=$1 stemgap=$2', round(d['ms_per_step'],3))" | tee -a $O/ab.log
  done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 5 --warmup 2 --no-legs --keep 0.6066 > $R/$O/prof.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_c/*.db | head -1) 24 "naive_conv|igemm_|Cijk" > $R/$O/kernel_stats.txt 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $R/$O/period.txt 2>&1
head -20 $R/$O/period.txt
