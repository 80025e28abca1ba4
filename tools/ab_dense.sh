set -x
B="python bench.py --steps 10 --warmup 3 --no-legs"
for i in 1 2; do
for w in channel spatial; do
  echo "== new $w"; $B --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS', d['ms_per_step'])"
  echo "== prev $w"; LDN_LIB_PATH=$PWD/tools/ablate/libldn_prev.so $B --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS', d['ms_per_step'])"
  echo "== nodense $w"; LDN_DENSE_KERNEL=0 $B --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS', d['ms_per_step'])"
done; done
