#!/bin/bash
# The rocprofv3 evidence for one round, each pass under its own timeout:
#   tools/profile_round.sh gpurun_out/prof_r1v5
# trace/     rocprofv3 --kernel-trace --stats  -- python bench.py      (per-kernel time)
# pmc_fetch/ rocprofv3 --pmc FETCH_SIZE --kernel-trace                  (HBM bytes read, own pass)
# pmc_write/ rocprofv3 --pmc WRITE_SIZE --kernel-trace                  (HBM bytes written, own pass)
# then: python profiles/summarize.py <outdir> profiles/<name> --kernel decode_fast
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$1
mkdir -p $OUT
timeout 280 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $OUT/bench_under_trace.json 2> $OUT/trace.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o r -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 > /dev/null 2> $OUT/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o r -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 > /dev/null 2> $OUT/pmc_write.err
timeout 200 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.db" | head; tail -c 600 $OUT/bench.json
