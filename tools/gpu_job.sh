#!/bin/bash
# One parametrised GPU job (replaces the one-off run_r*.sh scripts of earlier rounds).  Runs ON the GPU box, from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/gpu_job.sh <out-tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<out-tag>/ (merged back by gpurun).  Steps:
#   tests[:<pytest -k expr>]          pytest -m gpu (optionally filtered)           -> tests.log
#   testfile:<path>[::k]              one test file                                  -> tests_<name>.log
#   bench[:<workload>[:extra args]]   bench.py --brief of a workload (default: all legs of the headline when workload = full)
#   trace:<workload>[:<keep>]         rocprofv3 --kernel-trace --stats over bench.py --no-legs  -> <w>_kernel_stats.txt, period_<w>.txt
#   pmc:<workload>:<kernel-substr>[:<keep>]  MFMA-busy / traffic counters of the matching kernels (separate passes)  -> pmc_<w>.txt
#   py:<script and args>              python <script> (tools/*.py micro-benchmarks)  -> py_<n>.log
#   env:NAME=VALUE                    export for the following steps
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
EXCL="naive_conv|igemm_|grouped_conv_fwd|SubTensorOp|Im2d2Col|Cijk"
n=0
keep_of() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["config"].get("keep_probability_calibrated_to")
    print("--keep %r" % k if k is not None else "")
except Exception:
    print("")
PY
}
for step in "$@"; do
  n=$((n + 1))
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  echo "=== [$n] $step  $(date +%T)"
  case $kind in
    env) export "$rest" ;;
    tests)
      cd $R
      if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$rest" > $OUT/tests.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; fi
      echo "rc $?" >> $OUT/tests.log; tail -5 $OUT/tests.log ;;
    testfile)
      cd $R
      f=${rest%%::*}; k=${rest#*::}; [ "$k" = "$rest" ] && k=""
      name=$(basename $f .py)
      if [ -n "$k" ]; then timeout 1500 python -m pytest $f -m gpu -x -q -k "$k" > $OUT/tests_$name.log 2>&1
      else timeout 1500 python -m pytest $f -m gpu -x -q > $OUT/tests_$name.log 2>&1; fi
      echo "rc $?" >> $OUT/tests_$name.log; tail -5 $OUT/tests_$name.log ;;
    bench)
      cd $R
      w=${rest%%:*}; extra=${rest#*:}; [ "$extra" = "$rest" ] && extra=""
      [ -z "$w" ] && w=full
      if [ "$w" = full ]; then timeout 1200 python bench.py $extra > $OUT/bench_headline.json 2> $OUT/bench_headline.err
      else timeout 900 python bench.py --workload $w --brief $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err; fi
      python - $OUT/bench_$([ "$w" = full ] && echo headline || echo $w).json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print({k: d.get(k) for k in ("ms_per_step", "value", "realised_speedup_vs_dense_emulation", "vs_baseline")},
          {k: r.get(k) for k in ("frac", "avg_launch_us", "avg_us_per_block")})
    r3 = d.get("roofline_rows_3x3") or {}
    if r3: print("rows_3x3", {k: r3.get(k) for k in ("frac", "frac_executed", "avg_launch_us", "launches", "timed_ms_per_step")})
    for w, s in (d.get("secondary") or {}).items():
        print(w, {k: s.get(k) for k in ("ms_per_step", "realised_speedup_vs_dense_emulation", "max_abs_diff_vs_oracle_same_masks")})
except Exception as e:
    print("no JSON line:", e)
PY
      ;;
    trace)
      w=${rest%%:*}; kk=${rest#*:}; [ "$kk" = "$rest" ] && kk=""
      steps=3; [ $w = channel ] && steps=5
      cd /tmp; rm -rf /tmp/prof_$w
      bj=$OUT/bench_$w.json; [ $w = channel ] && bj=$OUT/bench_headline.json
      KEEP=$(keep_of $bj); [ -n "$kk" ] && KEEP="--keep $kk"
      echo "profiled command: bench.py --workload $w --steps $steps --warmup 2 --no-legs $KEEP" > $OUT/prof_$w.cmd
      # LDN_BENCH_NO_EVENTS: no event-bracketed roofline leg in the profiled run -- its HIP event records show up as ~5.6 us "gaps" behind every
      # bracketed launch (round 5: that is what the "unexplained gap" of earlier rounds was); the trace then holds uninstrumented forwards only
      LDN_BENCH_NO_EVENTS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o r -- python $R/bench.py --workload $w --steps $steps --warmup 2 --no-legs $KEEP > $OUT/prof_$w.log 2>&1
      python $R/tools/rocpd_stats.py $(ls /tmp/prof_$w/*.db | head -1) 30 "$EXCL" > $OUT/${w}_kernel_stats.txt 2>&1
      [ $w != adavit ] && python $R/tools/rocpd_period.py $(ls /tmp/prof_$w/*.db | head -1) 15 > $OUT/period_$w.txt 2>&1
      head -14 $OUT/${w}_kernel_stats.txt ;;
    pmc)
      w=${rest%%:*}; pat=${rest#*:}; kk=${pat#*:}; [ "$kk" = "$pat" ] && kk=""; pat=${pat%%:*}
      cd /tmp; rm -f $OUT/pmc_$w.txt
      bj=$OUT/bench_$w.json; [ $w = channel ] && bj=$OUT/bench_headline.json
      KEEP=$(keep_of $bj); [ -n "$kk" ] && KEEP="--keep $kk"
      for c in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_x
        LDN_BENCH_NO_EVENTS=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_x -o r -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-legs $KEEP > /tmp/pmc_x.log 2>&1
        echo "== $c (per dispatch; columns in alphabetical order of the counter names; FETCH_SIZE / WRITE_SIZE in KiB)" >> $OUT/pmc_$w.txt
        python $R/tools/rocpd_pmc.py $(ls /tmp/pmc_x/*.db | head -1) "$pat" 2>&1 | { read h; echo "$h"; tail -8; } >> $OUT/pmc_$w.txt
      done
      cat $OUT/pmc_$w.txt ;;
    py)
      cd $R
      timeout 900 python $rest > $OUT/py_$n.log 2>&1; echo "rc $?" >> $OUT/py_$n.log; tail -30 $OUT/py_$n.log ;;
    *) echo "unknown step $step" ;;
  esac
done
ls -la $OUT
