#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4p5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/plan_times.txt
for cfg in "14 56 256" "7 28 512" "7 14 1024" "7 7 2048"; do
  for mode in one pm decide two; do
    [ "$cfg" = "7 7 2048" ] && [ $mode != one ] && [ $mode != two ] && continue
    rm -rf /tmp/pt
    rocprofv3 --kernel-trace --stats -d /tmp/pt -o r -- python $R/tools/experiments/time_plan1.py $cfg $mode > /dev/null 2>&1
    echo "== $cfg $mode" >> $OUT/plan_times.txt
    python $R/tools/rocpd_stats.py $(ls /tmp/pt/*.db | head -1) 8 | grep -E "k_plan|k_mask|k_zero" >> $OUT/plan_times.txt
  done
done
cat $OUT/plan_times.txt
