#!/bin/bash
# one guarded rocprofv3 --pmc pass over an arbitrary command:
#   tools/pmc_cmd.sh <outdir> "<counters>" <kernel-substring> <command...>
export TMPDIR=/tmp
OUT=$1; CTRS=$2; PAT=$3; shift; shift; shift
mkdir -p $OUT
timeout 150 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o p -- "$@" > $OUT/out.txt 2> $OUT/err.txt
python tools/pmc_report.py $OUT $PAT
