#!/bin/bash
# one guarded rocprofv3 --pmc pass over bench.py: tools/pmc_quick.sh <outdir> "<counters>" <kernel-substring> [bench args]
export TMPDIR=/tmp
OUT=$1; CTRS=$2; PAT=$3; shift; shift; shift
mkdir -p $OUT
timeout 150 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --ramp-ms 0 "$@" > $OUT/bench.json 2> $OUT/err.txt
python tools/pmc_report.py $OUT $PAT
