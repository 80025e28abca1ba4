R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3i
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_blocks.py -x -q -m gpu -k "regnet" > $OUT/t_regnet.log 2>&1; echo "regnet tests rc=$?"; tail -n 3 $OUT/t_regnet.log
LDN_STAGE_CHUNK=16 timeout 1200 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "channel" > $OUT/t_chunk.log 2>&1; echo "chunked fullsize rc=$?"; tail -n 3 $OUT/t_chunk.log
for c in 0 64 128 32 0 64; do
LDN_STAGE_CHUNK=$c timeout 600 python bench.py --steps 10 --warmup 3 --no-legs > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err
python -c "
import json
d=json.loads(open('$OUT/bench_c$c.json').read().strip().splitlines()[-1]); print('chunk $c', round(d['ms_per_step'],3), round(d['value']))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
LDN_STAGE_CHUNK=64 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 3 --warmup 2 --no-legs > $OUT/prof.log 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $OUT/period_chunk64.txt 2>&1
head -50 $OUT/period_chunk64.txt | cut -c1-110
