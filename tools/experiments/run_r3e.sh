R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_d
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_d -o r -- python $R/tools/exp_dense.py 50176 2048 1024 > /tmp/pmc_d.log 2>&1
  echo "== $c" >> $OUT/pmc_dense.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_d/*.db | head -1) k_dense 2>&1 | tail -4 >> $OUT/pmc_dense.txt
done
cat $OUT/pmc_dense.txt
