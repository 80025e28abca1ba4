R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 0.62 0.75; do
rm -rf /tmp/prof_l
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o r -- python $R/bench.py --workload layer --steps 3 --warmup 2 --no-legs --keep $k > $OUT/prof_$k.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_l/*.db | head -1) 25 "naive_conv|igemm_|Cijk" > $OUT/stats_layer_$k.txt 2>&1
done
paste -d'\n' $OUT/stats_layer_0.62.txt | cut -c1-150 | head -30
echo =====
cut -c1-150 $OUT/stats_layer_0.75.txt | head -30
