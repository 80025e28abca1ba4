#!/bin/bash
# Round-2 rocprofv3 evidence (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_round2.sh').
# Raw databases stay in /tmp; text summaries go to gpurun_out/ (copy the ones to keep into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXCL="naive_conv|igemm_|grouped_conv_fwd|SubTensorOp|Im2d2Col|Cijk"
# 1. kernel-trace summaries of the bench workloads (same command line as the bench, product path only)
for w in ${WORKLOADS:-channel spatial layer regnet}; do
  steps=3; [ $w = channel ] && steps=5
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o r -- python $R/bench.py --workload $w --steps $steps --warmup 2 --no-legs > $OUT/r2_bench_$w.log 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_$w/*.db | head -1) 30 "$EXCL" > $OUT/r2_stats_$w.txt 2>&1
  [ $w = channel ] && python $R/tools/rocpd_timeline.py $(ls /tmp/prof_$w/*.db | head -1) 15 > $OUT/r2_timeline_channel.txt 2>&1
done
[ -n "$ONLY_STATS" ] && exit 0
# 2. PMC passes (one counter set per run, kernel-trace only) over the stage-3 launch of the fused tail (tools/trace_tail.py 3)
rm -f $OUT/r2_pmc_tail.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_t
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_t -o r -- python $R/tools/trace_tail.py 3 > /tmp/pmc_t.log 2>&1
  echo "== $c   (FETCH_SIZE / WRITE_SIZE in KiB per dispatch)" >> $OUT/r2_pmc_tail.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_t/*.db | head -1) k_tail 2>&1 | awk 'NR==1 || !seen[$2$3$4$5$6$7$8]++' | tail -4 >> $OUT/r2_pmc_tail.txt
done
# 3. the same PMC passes over the chained stage-3 launch (k_chain: 22 blocks of LAUD-ResNet101 in one launch) inside the bench itself
rm -f $OUT/r2_pmc_chain.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc_c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_c -o r -- python $R/bench.py --steps 2 --warmup 1 --no-legs > /tmp/pmc_c.log 2>&1
  echo "== $c   (FETCH_SIZE / WRITE_SIZE in KiB per dispatch)" >> $OUT/r2_pmc_chain.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_c/*.db | head -1) k_chain 2>&1 | tail -3 >> $OUT/r2_pmc_chain.txt
done
python $R/tools/rocpd_period.py $(ls /tmp/prof_channel/*.db | head -1) 15 > $OUT/r2_period_channel.txt 2>&1
