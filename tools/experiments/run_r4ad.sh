#!/bin/bash
run() { timeout 300 python bench.py "$@" --no-legs --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for w in spatial layer regnet channel; do run --workload $w; run --workload $w --graph; done
