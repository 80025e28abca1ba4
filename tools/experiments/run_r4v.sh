#!/bin/bash
for lib in trace wbursttrace; do
  echo "== $lib"
  KEEP=0.6066 LDN_LIB_PATH=tools/ablate/libldn_$lib.so timeout 300 python tools/trace_chain.py 2>&1 | grep -v amdgpu.ids
  for s in 1 2; do LDN_LIB_PATH=tools/ablate/libldn_$lib.so timeout 300 python tools/trace_head.py $s 2>&1 | grep -E "launch us"; done
done
run() { env "$@" timeout 300 python bench.py --no-legs --steps 30 --warmup 5 --keep 0.6066 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2 3 4; do run A=wlate; run LDN_LIB_PATH=tools/ablate/libldn_wburst.so; done
