R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_adavit.py -x -q -m gpu > $OUT/t_ada.log 2>&1; echo "adavit rc=$?"; tail -n 5 $OUT/t_ada.log
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "conv_rows or wide_1x1 or dense" > $OUT/t_ops.log 2>&1; echo "ops rc=$?"; tail -n 3 $OUT/t_ops.log
timeout 600 python bench.py --workload adavit > $OUT/bench_adavit.json 2> $OUT/bench_adavit.err; python -c "
import json
d=json.loads(open('$OUT/bench_adavit.json').read().strip().splitlines()[-1]); print('adavit', round(d['ms_per_step'],3), d['realised_speedup_vs_dense_emulation'], d['dense_emulation_gpu']['max_abs_diff_vs_hip_same_masks'], d['roofline']['bound'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_a
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o r -- python $R/bench.py --workload adavit --steps 3 --warmup 2 > $OUT/prof.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_a/*.db | head -1) 16 > $OUT/adavit_stats.txt 2>&1; cut -c1-150 $OUT/adavit_stats.txt
