#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4p10; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_plan.py -x -q -m gpu -k layer 2>&1 | tail -2
for v in 1 0 1 0; do
LDN_FUSED_SPATIAL_MASKER=$v timeout 600 python bench.py --workload layer --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layer fused=$v', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl; rocprofv3 --kernel-trace --stats -d /tmp/pl -o r -- python $R/bench.py --workload layer --steps 3 --warmup 2 --no-legs --keep 0.5 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/pl/*.db | head -1) 16 "naive_conv|igemm_|Cijk|ck::|_ZN2ck|SubTensor" | cut -c1-60,90-160 | tee $OUT/stats_layer_on.txt
python $R/tools/rocpd_period.py $(ls /tmp/pl/*.db | head -1) 15 > $OUT/period_layer_on.txt 2>&1
