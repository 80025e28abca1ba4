#!/bin/bash
# Round-4 rocprofv3 evidence (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_round4.sh').
# Raw databases stay in /tmp; text summaries go to gpurun_out/r4prof/ (copy the ones to keep into profiles/ as r04_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXCL="naive_conv|igemm_|grouped_conv_fwd|SubTensorOp|Im2d2Col|Cijk"
# 0. the bench lines themselves (all legs) -- same box, same call as the profiles below (ONLY_TRACE=1 skips to step 1 and reuses them)
cd $R
if [ -z "$ONLY_TRACE" ]; then
timeout 900 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err
for w in ${WORKLOADS:-spatial layer regnet adavit}; do
  timeout 900 python bench.py --workload $w --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
timeout 900 python bench.py --keep 0.5 --no-cpu > $OUT/bench_keep05.json 2> $OUT/bench_keep05.err
fi
cd /tmp
# the keep probability the bench line's bisection arrived at (seeded: the same every run): the profiled command passes it with --keep, so
# that the trace holds the warm-up + timed forwards only and the per-kernel averages are those of the bench line's workload
keep_of() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["config"].get("keep_probability_calibrated_to")
    print("--keep %r" % k if k is not None else "")
except Exception:
    print("")
PY
}
# 1. kernel-trace summaries of the bench workloads (same command line as the bench, product path only)
for w in channel ${WORKLOADS:-spatial layer regnet adavit}; do
  steps=3; [ $w = channel ] && steps=5
  rm -rf /tmp/prof_$w
  bj=$OUT/bench_$w.json; [ $w = channel ] && bj=$OUT/bench_headline.json
  [ -s $bj ] || { bj=$R/profiles/r04_bench_$w.json; [ $w = channel ] && bj=$R/profiles/r04_bench_headline.json; }     # ONLY_TRACE on a fresh box: the committed lines
  KEEP=$(keep_of $bj)
  echo "profiled command: bench.py --workload $w --steps $steps --warmup 2 --no-legs $KEEP" > $OUT/prof_$w.cmd
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o r -- python $R/bench.py --workload $w --steps $steps --warmup 2 --no-legs $KEEP > $OUT/prof_$w.log 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_$w/*.db | head -1) 30 "$EXCL" > $OUT/${w}_kernel_stats.txt 2>&1
  [ $w != adavit ] && python $R/tools/rocpd_period.py $(ls /tmp/prof_$w/*.db | head -1) 15 > $OUT/period_$w.txt 2>&1
done
[ -n "$ONLY_TRACE" ] && { ls -la $OUT; exit 0; }
# 2. PMC passes (one counter set per run, kernel-trace only) over the chained stage-3 launch inside the bench itself
rm -f $OUT/pmc_chain.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_c -o r -- python $R/bench.py --steps 2 --warmup 1 --no-legs $(keep_of $OUT/bench_headline.json) > /tmp/pmc_c.log 2>&1
  echo "== $c   (FETCH_SIZE / WRITE_SIZE in KiB per dispatch; columns in alphabetical order of the counter names)" >> $OUT/pmc_chain.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_c/*.db | head -1) k_chain 2>&1 | tail -3 >> $OUT/pmc_chain.txt
done
python $R/tools/make_traffic_json.py $OUT/pmc_chain.txt $OUT/traffic.json "$(date -u +%Y-%m-%d)"
# 3. PMC passes over the packed 3x3 of the spatial workload (k_dense<.., T9>): MFMA utilisation from counters over the launches
rm -f $OUT/pmc_rows3x3.txt
for c in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_s
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_s -o r -- python $R/bench.py --workload spatial --steps 2 --warmup 1 --no-legs $(keep_of $OUT/bench_spatial.json) > /tmp/pmc_s.log 2>&1
  echo "== $c (per dispatch; columns in alphabetical order)" >> $OUT/pmc_rows3x3.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_s/*.db | head -1) "true, true" 2>&1 | tail -8 >> $OUT/pmc_rows3x3.txt
done
ls -la $OUT
