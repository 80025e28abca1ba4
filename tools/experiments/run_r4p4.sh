#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4p4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KEEP="--keep 0.5"
for tag in ${TAGS:-on}; do
  rm -rf /tmp/prof_$tag
  if [ $tag = off ]; then export LDN_FUSED_SPATIAL_MASKER=0 LDN_INDEX_PLAN=0; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python $R/bench.py --workload spatial --steps 3 --warmup 2 --no-legs $KEEP > $OUT/prof_$tag.log 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_$tag/*.db | head -1) 14 "naive_conv|igemm_|Cijk" > $OUT/stats_$tag.txt 2>&1
  python $R/tools/rocpd_period.py $(ls /tmp/prof_$tag/*.db | head -1) 15 > $OUT/period_$tag.txt 2>&1
done
head -18 $OUT/stats_on.txt; head -18 $OUT/stats_off.txt
