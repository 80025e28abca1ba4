#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_stem.py -x -q -m gpu 2>&1 | tail -2
python tools/experiments/time_stem.py 2>&1 | grep "us per" | tee $O/stem_ablate3.log
for i in 1 2; do timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', round(d['ms_per_step'],3))"; done
