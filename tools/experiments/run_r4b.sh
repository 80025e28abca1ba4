#!/bin/bash
# round 4, call b: folded projection -- parity, A/B, and phase traces of the tails at stages 1-3
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_tail.py tests/test_hip_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
for i in 1 2; do
  for f in 0 1; do
    LDN_FOLD_PROJ=$f timeout 300 python bench.py --no-legs --steps 20 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=$f', round(d['ms_per_step'],3))" | tee -a $O/ab.log
  done
done
for s in 1 2 3; do LDN_LIB_PATH=tools/ablate/libldn_trace.so timeout 300 python tools/trace_tail.py $s > $O/trace_tail_$s.log 2>&1; done
cat $O/trace_tail_*.log
