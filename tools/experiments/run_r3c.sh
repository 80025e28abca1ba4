R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
cd $R
LDN_DENSE16=3 timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dense16" > $OUT/t_ops3.log 2>&1; echo "ops tests (K16) rc=$?"; tail -n 3 $OUT/t_ops3.log
LDN_DENSE16=1 timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dense16" > $OUT/t_ops1.log 2>&1; echo "ops tests (16 waves) rc=$?"; tail -n 3 $OUT/t_ops1.log
for v in 0 3 0 3; do
LDN_DENSE16=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-legs > $OUT/bench_d$v.json 2> $OUT/bench_d$v.err; echo "bench d$v rc=$?"
python -c "
import json
d=json.loads(open('$OUT/bench_d$v.json').read().strip().splitlines()[-1]); print('channel d$v', d['ms_per_step'], d['value'])"
done
for v in 0 3; do
LDN_DENSE16=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --workload spatial > $OUT/bench_sp$v.json 2> $OUT/bench_sp$v.err
python -c "
import json
d=json.loads(open('$OUT/bench_sp$v.json').read().strip().splitlines()[-1]); print('spatial d$v', d['ms_per_step'], d['value'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
LDN_DENSE16=3 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 3 --warmup 2 --no-legs > $OUT/prof.log 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $OUT/period_channel_k16.txt 2>&1
grep "k_dense" $OUT/period_channel_k16.txt
