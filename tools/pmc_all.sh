#!/bin/bash
# tuning / audit: FETCH_SIZE and WRITE_SIZE (KiB; FETCH x 2 on gfx950) of EVERY library kernel of a workload, averaged per (kernel, grid) -- the table to compare with the bytes each
# launch must touch ("traffic well above the algorithmic bytes = wasted re-reads").   usage (on the GPU box): tools/pmc_all.sh <out-dir> <workload>[:<keep>] ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift; case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for spec in "$@"; do
  w=${spec%%:*}; k=${spec#*:}; [ "$k" = "$spec" ] && k=""
  KEEP=""; [ -n "$k" ] && KEEP="--keep $k"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_all
    LDN_BENCH_NO_EVENTS=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_all -o r -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-legs $KEEP > /tmp/pmc_all.log 2>&1
    echo "== $w $c (KiB per launch, averaged per kernel and grid)" >> $OUT/pmc_all_$w.txt
    python $R/tools/rocpd_pmc_avg.py $(ls /tmp/pmc_all/*.db | head -1) "ldn::" >> $OUT/pmc_all_$w.txt 2>&1
  done
done
