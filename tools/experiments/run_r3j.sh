R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3j
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_tail.py tests/test_hip_chain.py tests/test_hip_debug.py -x -q -m gpu > $OUT/t_tail.log 2>&1; echo "tail tests rc=$?"; tail -n 3 $OUT/t_tail.log
timeout 1200 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "channel" > $OUT/t_full.log 2>&1; echo "fullsize rc=$?"; tail -n 3 $OUT/t_full.log
timeout 600 python -m pytest tests/test_hip_adavit.py -x -q -m gpu > $OUT/t_ada.log 2>&1; echo "adavit rc=$?"; tail -n 3 $OUT/t_ada.log
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-legs > $OUT/bench_$i.json 2> $OUT/bench_$i.err
python -c "
import json
d=json.loads(open('$OUT/bench_$i.json').read().strip().splitlines()[-1]); print('run $i', round(d['ms_per_step'],3), round(d['value']))"
done
timeout 600 python bench.py --workload adavit > $OUT/bench_adavit.json 2> $OUT/bench_adavit.err; python -c "
import json
d=json.loads(open('$OUT/bench_adavit.json').read().strip().splitlines()[-1]); print('adavit', round(d['ms_per_step'],3), d['realised_speedup_vs_dense_emulation'], d['dense_emulation_gpu']['max_abs_diff_vs_hip_same_masks'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r -- python $R/bench.py --steps 3 --warmup 2 --no-legs > $OUT/prof.log 2>&1
python $R/tools/rocpd_period.py $(ls /tmp/prof_c/*.db | head -1) 15 > $OUT/period_channel.txt 2>&1
grep "k_tail<\|k_head<\|k_chain\|period" $OUT/period_channel.txt | cut -c1-100
