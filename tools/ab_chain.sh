#!/bin/bash
# tuning only: same-box A/B of library variants on the headline workload.  usage (on the GPU box): tools/ab_chain.sh <rounds> <lib> [<lib> ...]
# every lib is a path relative to the repo root ("default" = laudnet_amd/libldn_hip.so); prints ms per step and us per chained block, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset LDN_LIB_PATH; else export LDN_LIB_PATH=$R/$lib; fi
    python $R/bench.py --workload channel --brief --steps 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline') or {}
print('$lib', 'round $r', 'ms/step %.3f' % d['ms_per_step'], 'us/block %.1f' % r.get('avg_us_per_block', 0), 'frac %.4f' % r.get('frac', 0))"
  done
done
