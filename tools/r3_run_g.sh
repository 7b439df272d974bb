cd $GRAFT_REPO_ROOT
for v in xp32 xp32lean; do echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_data.py tests/test_gpu_norle.py tests/test_gpu_query.py -m gpu -x -q 2>&1 | tail -3; done
python tools/ab.py --cfg headline --rounds 4 base=sprintz_amd/variants/base.so xp32=sprintz_amd/variants/xp32.so xp32lean=sprintz_amd/variants/xp32lean.so 2>&1 | tail -16
