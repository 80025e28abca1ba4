#!/bin/bash
# Regenerates the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root:
#   gpurun -- 'bash tools/profile_round.sh').  Raw databases stay in /tmp; only text summaries go to gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXCL="naive_conv|igemm_|grouped_conv_fwd|SubTensorOp|Im2d2Col|Cijk"
# 1. kernel-trace summaries of the bench workloads (headline: 5 timed steps, others: 3)
for w in channel spatial layer regnet; do
  steps=3; [ $w = channel ] && steps=5
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o r -- python $R/bench.py --workload $w --steps $steps --warmup 2 --no-legs > $OUT/bench_$w.log 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_$w/*.db | head -1) 30 "$EXCL" > $OUT/stats_$w.txt 2>&1
done
[ -n "$ONLY_STATS" ] && exit 0
rm -f $OUT/pmc_traffic.txt $OUT/pmc_sq.txt
# 2. PMC passes (separate runs per counter, kernel-trace only) over the stage-3 instances of the two dominant kernels
for m in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$m$c
    LDN_MATH_MODE=$m timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$m$c -o r -- python $R/tools/bench_conv.py --stage 3 --kinds conv2,conv3 --iters 3 > /tmp/pmc_$m$c.log 2>&1
    echo "== math_mode=$m $c (KiB per dispatch)" >> $OUT/pmc_traffic.txt
    python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_$m$c/*.db | head -1) k_conv 2>&1 | tail -10 >> $OUT/pmc_traffic.txt
  done
done
# 3. matrix-pipe / LDS counters of the same launches in the headline math mode
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_sq
  LDN_MATH_MODE=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_sq -o r -- python $R/tools/bench_conv.py --stage 3 --kinds conv2,conv3 --iters 3 > /tmp/pmc_sq.log 2>&1
  echo "== $set" >> $OUT/pmc_sq.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_sq/*.db | head -1) k_conv 2>&1 | awk 'NR==1 || !seen[$2$3$4$5$6$7$8]++' >> $OUT/pmc_sq.txt
done
