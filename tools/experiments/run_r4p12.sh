#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4p12; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_adavit.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --workload adavit --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adavit', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3), d.get('dense_emulation_gpu',{}).get('max_abs_diff_vs_hip_same_masks'))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa; rocprofv3 --kernel-trace --stats -d /tmp/pa -o r -- python $R/bench.py --workload adavit --steps 3 --warmup 2 --no-legs > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/pa/*.db | head -1) 16 | cut -c1-75,90-160 | tee $OUT/stats_adavit.txt
