R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3m
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_hip_blocks.py tests/test_hip_ops.py tests/test_hip_fullsize.py -x -q -m gpu > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -n 4 $OUT/t.log
for v in 1 0 1 0; do
LDN_LAYER_CARRY=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --workload spatial > $OUT/bench_sp$v.json 2> $OUT/bench_sp$v.err
python -c "
import json
d=json.loads(open('$OUT/bench_sp$v.json').read().strip().splitlines()[-1]); print('spatial carry $v', round(d['ms_per_step'],3), round(d['value']), d['config']['mean_block_flops_ratio'])"
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --workload layer > $OUT/bench_layer.json 2> $OUT/bench_layer.err
python -c "
import json
d=json.loads(open('$OUT/bench_layer.json').read().strip().splitlines()[-1]); print('layer', round(d['ms_per_step'],3), round(d['value']), d['config']['mean_block_flops_ratio'])"
