#!/bin/bash
# texture-addresser / L1 counters of one bench configuration's kernels: tools/ta_busy.sh <outdir under gpurun_out> <config> <kernel substring> [extra bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$1; CFG=$2; PAT=$3; shift; shift; shift
mkdir -p $OUT
if [ "$CFG" = headline ]; then ARGS="--configs none --no-extras --no-sweep --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0"; else ARGS="--only $CFG --no-cpu-baseline --config-reps 3 --no-extras"; fi
(cd /tmp && timeout 200 rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ta -o p -- python $ROOT/bench.py $ARGS "$@" > /dev/null 2> $OUT/ta.err < /dev/null)
python tools/pmc_report.py $OUT/ta $PAT | tee $OUT/ta_report.txt
