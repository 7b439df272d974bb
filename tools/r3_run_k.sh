cd $GRAFT_REPO_ROOT
echo "== tests with spec"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/spec.so python -m pytest tests/test_gpu_huf0.py tests/test_gpu_bench_data.py -m gpu -x -q 2>&1 | tail -3
for c in cfg4_800000 cfg4_10000; do python tools/ab.py --cfg $c --rounds 3 --reps 8 nospec=sprintz_amd/variants/nospec.so spec=sprintz_amd/variants/spec.so 2>&1 | tail -8; done
