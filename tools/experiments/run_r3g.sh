R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3g
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $OUT/t_all.log
timeout 900 python bench.py > $OUT/bench_head.json 2> $OUT/bench_head.err; echo "bench rc=$?"
LDN_FUSED_STATS=0 timeout 600 python bench.py --no-legs > $OUT/bench_nostats.json 2> $OUT/bench_nostats.err
timeout 600 python bench.py --no-legs > $OUT/bench_stats.json 2> $OUT/bench_stats.err
timeout 600 python bench.py --workload regnet --no-cpu > $OUT/bench_regnet.json 2> $OUT/bench_regnet.err; echo "regnet rc=$?"
timeout 600 python bench.py --workload adavit > $OUT/bench_adavit.json 2> $OUT/bench_adavit.err; echo "adavit rc=$?"
python - <<PY
import json
for f in ('bench_head','bench_nostats','bench_stats','bench_regnet','bench_adavit'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), round(d['value']), 'x', d.get('realised_speedup_vs_dense_emulation'), 'roof', (d.get('roofline') or {}).get('bound'), (d.get('roofline') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -n 5 $OUT/bench_head.err
