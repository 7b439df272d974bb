#!/bin/bash
# The rocprofv3 evidence of round 6, each pass under its own timeout, one sub-directory per workload:
#   tools/profile_round6.sh <outdir under gpurun_out> [workloads...]     (default: headline + every per_config entry)
# <w>/trace      rocprofv3 --kernel-trace --stats      (per-kernel time)
# <w>/pmc_fetch  rocprofv3 --pmc FETCH_SIZE            (HBM bytes read; own pass, --kernel-trace only)
# <w>/pmc_write  rocprofv3 --pmc WRITE_SIZE            (HBM bytes written; own pass)
# then, in the build container:  python profiles/summarize_r2.py gpurun_out/<outdir> profiles/r2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$1; shift
WL=${@:-headline cfg1 cfg3_1k cfg3_10k cfg4_1250 cfg4_10000 cfg4_80000 cfg4_800000 cfg5 cfg5_8m}
mkdir -p $OUT
# the default bench run first: the record the contract test reads (bench_full.json) and the line the driver would parse
[ -n "$SKIP_BENCH" ] || (timeout 900 python $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt < /dev/null; cp $ROOT/bench_full.json $OUT/bench_full.json 2>/dev/null; echo "bench rc $? $(date +%T)")
for w in $WL; do
  if [ $w = headline ]; then ARGS="--configs none --no-extras --no-sweep --no-cpu-baseline --steps 20 --warmup 3"; PARGS="--configs none --no-extras --no-sweep --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0"
  else ARGS="--only $w --no-cpu-baseline --config-reps 20"; PARGS="--only $w --no-cpu-baseline --config-reps 3"; fi
  mkdir -p $OUT/$w
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/$w/trace -o r -- python $ROOT/bench.py $ARGS > $OUT/$w/bench_under_trace.json 2> $OUT/$w/trace.err < /dev/null)
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/$w/pmc_fetch -o r -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/$w/pmc_fetch.err < /dev/null)
  (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/$w/pmc_write -o r -- python $ROOT/bench.py $PARGS > /dev/null 2> $OUT/$w/pmc_write.err < /dev/null)
  echo "$w done $(date +%T)"
done
# the databases are tens of MB each: summarise here, carry only the text back (gpurun merges <= 64 MiB)
mkdir -p $OUT/summary
python profiles/summarize_r2.py $OUT $OUT/summary/r6 --build "${BUILD_ID:-unlabelled}" --traffic-json $OUT/summary/hbm_traffic.json > $OUT/summary/summarize.log 2>&1
for w in $WL; do cp $OUT/$w/bench_under_trace.json $OUT/summary/${w}_bench_under_trace.json 2>/dev/null; rm -rf $OUT/$w; done
ls $OUT/summary | head -40
