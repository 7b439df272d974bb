#!/bin/bash
# like ktrace.sh but no verification (truncated/ablated kernels): tools/ktrace2.sh <out> <cfg> <pattern> lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; CFG=$2; PAT=$3; shift; shift; shift
mkdir -p $OUT
for LIB in "$@"; do
  N=$(basename $LIB .so)
  (cd /tmp && SPRINTZ_MI355X_LIB=$GRAFT_REPO_ROOT/$LIB timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t_$N -o r -- python $GRAFT_REPO_ROOT/bench.py --only $CFG --no-cpu-baseline --no-verify --config-reps 10 > /dev/null 2> $OUT/$N.err)
  python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$OUT/t_$N/*.db")[0])
for r in db.execute("select name,total_calls,average from top_kernels"):
    if "$PAT" in r[0]: print(f"$N {r[0][:70]:70s} calls {r[1]:4d} avg {r[2]:9.1f} us")
PY
done
