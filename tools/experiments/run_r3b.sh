R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dense16 or conv_rows or wide_1x1 or dense_kernel" > $OUT/t_ops.log 2>&1; echo "ops tests rc=$?"; tail -n 3 $OUT/t_ops.log
for v in 0 1 2; do
LDN_DENSE16=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-legs > $OUT/bench_d$v.json 2> $OUT/bench_d$v.err; echo "bench d$v rc=$?"
done
LDN_DENSE16=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --workload spatial > $OUT/bench_sp1.json 2> $OUT/bench_sp1.err
LDN_DENSE16=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --workload spatial > $OUT/bench_sp0.json 2> $OUT/bench_sp0.err
LDN_DENSE16=2 timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --workload spatial > $OUT/bench_sp2.json 2> $OUT/bench_sp2.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 3 --warmup 2 --no-legs > $OUT/prof.log 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $OUT/period_channel.txt 2>&1
python -c "
import json
for f in ('bench_d0','bench_d1','bench_d2','bench_sp0','bench_sp1','bench_sp2'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
"
