#!/bin/bash
# tuning only: counters of k_smallmap (one counter set per pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sm; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/pmc.txt
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_LDS" GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_sm
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_sm -o r -- python $R/tools/bench_small.py 0.62 > /tmp/pmc_sm.log 2>&1
  echo "== $c" >> $OUT/pmc.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_sm/*.db | head -1) k_smallmap 2>&1 | tail -2 >> $OUT/pmc.txt
done
cat $OUT/pmc.txt
