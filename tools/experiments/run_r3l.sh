R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3l
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_tail.py -x -q -m gpu > $OUT/t_tail.log 2>&1; echo "tail tests rc=$?"; tail -n 8 $OUT/t_tail.log
timeout 1500 python -m pytest tests/test_hip_fullsize.py tests/test_hip_blocks.py tests/test_hip_chain.py -x -q -m gpu > $OUT/t_full.log 2>&1; echo "fullsize rc=$?"; tail -n 3 $OUT/t_full.log
for v in 1 0 1 0; do
LDN_FUSE_MASKER=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-legs > $OUT/bench_f$v.json 2> $OUT/bench_f$v.err
python -c "
import json
d=json.loads(open('$OUT/bench_f$v.json').read().strip().splitlines()[-1]); print('fuse $v', round(d['ms_per_step'],3), round(d['value']))"
done
