#!/bin/bash
# round-2 GPU session driver: every step under its own timeout, everything into gpurun_out/$1
#   tools/r2_gpu.sh <outdir> <steps...>     steps: tests bench probe trace pmc8
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for step in "$@"; do
  echo "=== $step $(date +%T)"
  case $step in
    tests)  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log ;;
    dropin) timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q > $OUT/dropin.log 2>&1; echo "dropin rc=$?"; tail -15 $OUT/dropin.log ;;
    bench)  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.err; head -c 3000 $OUT/bench.json ;;
    benchq) timeout 600 python bench.py --configs none --no-extras --no-cpu-baseline > $OUT/benchq.json 2> $OUT/benchq.err; echo "benchq rc=$?"; tail -c 600 $OUT/benchq.err; head -c 1200 $OUT/benchq.json ;;
    probe)  timeout 120 tools/probes/lds_read2_misaligned > $OUT/probe_read2.txt 2>&1; cat $OUT/probe_read2.txt ;;
    trace)  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$OUT/bench_under_trace.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err); echo "trace rc=$?"; tail -3 $OUT/trace.err ;;
    pmc8)   for cfg in cfg1 cfg3_10k cfg4_80000; do
              (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$cfg -o p -- python $GRAFT_REPO_ROOT/bench.py --only $cfg --no-cpu-baseline --config-reps 3 > $GRAFT_REPO_ROOT/$OUT/pmc_$cfg.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$cfg.err)
              python tools/pmc_report.py $OUT/pmc_$cfg _kernel | tee $OUT/pmc_$cfg.txt
            done ;;
  esac
done
echo "=== done $(date +%T)"
