cd /tmp && export TMPDIR=/tmp LDN_MATH_MODE=1
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o r -- python $R/tools/bench_conv.py --stage 3 --kinds conv2,conv3 --iters 3 > /tmp/pmc$i.log 2>&1
  echo "== $set" >> $R/gpurun_out/pmc.txt
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmc$i/*.db | head -1) k_conv_bf3  2>&1 | grep -v "^dispatch" | awk "{k=\$2\$3\$4\$5\$6\$7\$8; if (!(k in seen)) {seen[k]=1; print}}" >> $R/gpurun_out/pmc.txt
done
