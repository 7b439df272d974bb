#!/bin/bash
# HBM bytes of one kernel under ANY command (FETCH_SIZE and WRITE_SIZE in passes of their own, the guide's corrections applied by the reader):
#   tools/traffic_cmd.sh <outdir under gpurun_out> <kernel substring> <command...>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$1; PAT=$2; shift 2
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- "$@" > $OUT/$c.out 2> $OUT/$c.err < /dev/null)
done
python tools/pmc_report.py $OUT "$PAT" | tee $OUT/report.txt
find $OUT -name "*.csv" -size +1000k -delete
