R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_tail.py tests/test_hip_chain.py -x -q -m gpu > $OUT/t_tail.log 2>&1; echo "tail tests rc=$?" 
timeout 1200 python -m pytest tests/test_hip_fullsize.py tests/test_hip_blocks.py -x -q -m gpu > $OUT/t_full.log 2>&1; echo "fullsize rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 --no-legs > $OUT/bench_new.json 2> $OUT/bench_new.err; echo "bench rc=$?"
LDN_TAIL_STRIDE2=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-legs > $OUT/bench_old.json 2> $OUT/bench_old.err; echo "bench old rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 3 --warmup 2 --no-legs > $OUT/prof.log 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $OUT/period_channel.txt 2>&1
tail -3 $OUT/t_tail.log $OUT/t_full.log
python -c "
import json
for f in ('bench_new','bench_old'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
"
