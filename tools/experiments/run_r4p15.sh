#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for w in channel spatial regnet; do
for v in 0 1 0 1; do
HIP_FORCE_DEV_KERNARG=$v timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --brief 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w devkernarg=$v', round(d['ms_per_step'],3), round(d.get('realised_speedup_vs_dense_emulation') or 0,3))"
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pn; HIP_FORCE_DEV_KERNARG=1 rocprofv3 --kernel-trace --stats -d /tmp/pn -o r -- python $R/bench.py --workload spatial --steps 3 --warmup 2 --no-legs --keep 0.5 > /dev/null 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/pn/*.db | head -1) 15 | sed -n 1,12p
