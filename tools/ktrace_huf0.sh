#!/bin/bash
# per-kernel times of tools/bench_huf0.py (libzstd blocks: a tree per chunk): tools/ktrace_huf0.sh <outdir-name> <chunks> [lib.so]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; N=$2; LIB=$3
mkdir -p $OUT
[ -n "$LIB" ] && export SPRINTZ_MI355X_LIB=$PWD/$LIB
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_h$N -o r -- python $GRAFT_REPO_ROOT/tools/bench_huf0.py --chunks $N > $OUT/h$N.txt 2> $OUT/h$N.err)
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$OUT/trace_h$N/*.db")[0])
for r in db.execute("select name,total_calls,average,percentage from top_kernels"):
    if any(k in r[0] for k in ("sprintz","huf","compact","scan_")):
        print(f"{r[0][:100]:100s} calls {r[1]:5d} avg {r[2]/1000:10.3f} ms")
PY
rm -rf $OUT/trace_h$N
