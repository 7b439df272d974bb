#!/bin/bash
# HBM-side traffic of one bench configuration's kernels: tools/pmc_fetch.sh <outdir-name> <cfg> <kernel-substr> [lib.so]
# (FETCH_SIZE and WRITE_SIZE in separate passes, KiB; FETCH_SIZE x 2 on gfx950 -- MI355X_MICROARCH.md)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; CFG=$2; PAT=$3; LIB=$4
mkdir -p $OUT
[ -n "$LIB" ] && export SPRINTZ_MI355X_LIB=$PWD/$LIB
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --only $CFG --no-cpu-baseline --config-reps 2 > /dev/null 2> $OUT/$c.err)
done
python tools/pmc_report.py $OUT $PAT | awk '{print $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
