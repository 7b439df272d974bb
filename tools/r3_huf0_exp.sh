cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== probe lds_read2_misaligned"; (cd tools/probes && make -s lds_read2_misaligned 2>&1 | tail -2; ./lds_read2_misaligned)
VARS="$@"
python tools/ab.py --cfg cfg4_800000 --rounds 2 --reps 8 $(for v in $VARS; do echo -n "$v=sprintz_amd/variants/$v.so "; done) 2>&1 | tail -12
for v in $VARS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$GRAFT_REPO_ROOT/gpurun_out/r3_huf0_$v_$c
    (cd /tmp && SPRINTZ_MI355X_LIB=$GRAFT_REPO_ROOT/sprintz_amd/variants/$v.so timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_huf0_${v}_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --only cfg4_800000 --no-cpu-baseline --config-reps 3 > /dev/null 2>/dev/null)
    python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r3_huf0_${v}_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "huf0_stream" in r["Kernel_Name"] or "decode_fast" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
for k,vv in acc.items(): print("$v $c", k, "n=%d avg=%.1f MB" % (len(vv), sum(vv)/len(vv)*1024/1e6))
PY
  done
done
