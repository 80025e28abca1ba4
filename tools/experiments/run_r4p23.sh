#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --stats -d /tmp/pg -o r -- python $R/bench.py --workload spatial --steps 3 --warmup 2 --no-legs --keep 0.5 > /dev/null 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/pg/*.db | head -1) 0 | sed -n 1,60p
