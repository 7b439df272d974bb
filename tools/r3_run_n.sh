cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in g4 g5; do echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests/test_gpu_huf0.py tests/test_gpu_bench_data.py -m gpu -x -q 2>&1 | tail -2; done
python tools/ab.py --cfg cfg4_800000 --rounds 3 --reps 8 prev=sprintz_amd/variants/prev.so g4=sprintz_amd/variants/g4.so g5=sprintz_amd/variants/g5.so 2>&1 | grep -E "MEDIAN|huff0_decode" | awk '{print $1,$2,$3,$4,$5,$9}'
python tools/ab.py --cfg cfg4_80000 --rounds 3 --reps 10 prev=sprintz_amd/variants/prev.so g4=sprintz_amd/variants/g4.so 2>&1 | grep MEDIAN
for v in prev g4; do
  (cd /tmp && SPRINTZ_MI355X_LIB=$GRAFT_REPO_ROOT/sprintz_amd/variants/$v.so timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_fetch_$v -o p -- python $GRAFT_REPO_ROOT/bench.py --only cfg4_800000 --no-cpu-baseline --config-reps 3 > /dev/null 2>/dev/null)
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r3_fetch_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "huf0_stream" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,vv in acc.items(): print("$v FETCH_SIZE", k, "n=%d avg=%.1f MB raw" % (len(vv), sum(vv)/len(vv)*1024/1e6))
PY
done
