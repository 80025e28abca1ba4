#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQ_INSTS_.*FETCH|SQC_" | head -40
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM"; do
  rm -rf /tmp/pmc_i
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_i -o r -- python $R/bench.py --steps 2 --warmup 1 --no-legs --keep 0.6066 > /tmp/pmc_i.log 2>&1
  echo "== $c"
  python $R/tools/rocpd_pmc_avg.py $(ls /tmp/pmc_i/*.db | head -1) "k_chain,k_tail,k_head,k_smallmap,k_dense" 2>&1 | tail -22
done
