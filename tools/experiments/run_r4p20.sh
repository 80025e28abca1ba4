#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4p20; mkdir -p $OUT
cd $R
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4p20/bench_default.json").read().strip().splitlines()[-1])
print("headline", round(d["ms_per_step"],3), d.get("realised_speedup_vs_dense_emulation"), d["roofline"]["frac"], d["roofline"].get("traffic"))
for k,v in d.get("secondary",{}).items():
    print(k, v.get("ms_per_step"), v.get("realised_speedup_vs_dense_emulation"), v.get("max_abs_diff_vs_oracle_same_masks"), v.get("error"))
P
cd /tmp && export TMPDIR=/tmp
for w in spatial adavit; do
rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats -d /tmp/pf -o r -- python $R/bench.py --workload $w --steps 12 --warmup 4 --no-legs $( [ $w = spatial ] && echo --keep 0.5 ) > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/pf/*.db | head -1) 16 "naive_conv|igemm_|Cijk|ck::|_ZN2ck|SubTensor" > $OUT/stats_$w.txt 2>&1
done
head -12 $OUT/stats_spatial.txt | cut -c1-70,90-160
