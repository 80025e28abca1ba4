R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3n
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_tail.py -x -q -m gpu > $OUT/t_tail.log 2>&1; echo "tail tests rc=$?"; tail -n 3 $OUT/t_tail.log
for v in two one two one; do
lib=$R/laudnet_amd/libldn_hip.so; [ $v = one ] && lib=$R/tools/ablate/libldn_tail4one.so
LDN_LIB_PATH=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --keep 0.62 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
python -c "
import json
d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1]); print('tail4 $v', round(d['ms_per_step'],3), round(d['value']))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 3 --warmup 2 --no-legs --keep 0.62 > $OUT/prof.log 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $OUT/period_channel.txt 2>&1
grep "k_tail<4\|k_head<4" $OUT/period_channel.txt | cut -c1-100
