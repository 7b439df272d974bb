#!/bin/bash
# one counter pass of the big-batch Huff0 stream kernel per library build: tools/pmc_huf0_quick.sh variants/a.so variants/b.so ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
for L in "$@"; do
  OUT=$ROOT/gpurun_out/pmcq_$(basename $L .so); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && SPRINTZ_MI355X_LIB=$ROOT/sprintz_amd/$L timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p -o p -- python $ROOT/bench.py --only cfg4_${CHUNKS:-800000} --no-cpu-baseline --config-reps 3 --no-verify > $OUT/p.json 2> $OUT/p.err < /dev/null)
  echo "== $L"; python tools/pmc_report.py $OUT "huf0_stream_kernel<true" | awk '{print $(NF-2), $(NF-1), $NF}'
  python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "huf0_stream_kernel<true" in r["Kernel_Name"]]
    if d: print("kernel us (under counters): n=%d median %.1f" % (len(d), sorted(d)[len(d) // 2]))
PY
  find $OUT -name "*.csv" -size +1000k -delete; rm -rf $OUT/p/*/*.db 2>/dev/null
done
