#!/bin/bash
# per-kernel times of ANY command: tools/ktrace_cmd.sh <outdir-name> <command...>   (every pass under its own timeout; never reads stdin)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; shift
mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- "$@" > $OUT/cmd.out 2> $OUT/cmd.err < /dev/null)
tail -c 1500 $OUT/cmd.out
python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/trace/**/*.db", recursive=True)
if not f:
    print("no rocprofv3 database under $OUT/trace")
else:
    db=sqlite3.connect(f[0])
    with open("$OUT/kernel_stats.txt","w") as o:
        for r in db.execute("select name,total_calls,average,percentage from top_kernels"):
            line=f"{r[0][:110]:110s} calls {r[1]:6d} avg {r[2]/1000:10.2f} us  {r[3]:5.1f}%"
            print(line); o.write(line+"\n")
PY
rm -rf $OUT/trace
