#!/bin/bash
# Round-4 PMC passes over the stage-1/2 kernels of the headline workload: k_head<2>, k_tail<2,1>, k_head<4>,
# k_tail<4,1>, k_tail<4,2>, the projections (k_dense<8>), k_stem, k_smallmap.  One counter set per run (kernel-trace only).
# gpurun -- 'bash tools/profile_r4_stage12.sh'  ->  gpurun_out/r4prof/pmc_stage12.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KEEP=${KEEP:---keep 0.6066}
F=$OUT/pmc_stage12.txt
echo "# bench.py --steps 2 --warmup 1 --no-legs $KEEP; averages per (kernel, grid); FETCH_SIZE / WRITE_SIZE in KiB (FETCH x2 per the guide); SQ_* wave counters in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM in cycles" > $F
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_c -o r -- python $R/bench.py --steps 2 --warmup 1 --no-legs $KEEP > /tmp/pmc_c.log 2>&1
  echo "== $c" >> $F
  python $R/tools/rocpd_pmc_avg.py $(ls /tmp/pmc_c/*.db | head -1) "k_head,k_tail,k_dense,k_stem,k_smallmap,k_chain,k_conv_bf3" >> $F 2>&1
done
cat $F
