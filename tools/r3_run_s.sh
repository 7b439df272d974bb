cd $GRAFT_REPO_ROOT
echo "== tests with split"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/split.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_data.py tests/test_gpu_norle.py tests/test_gpu_query.py tests/test_gpu_colmajor.py -m gpu -x -q 2>&1 | tail -2
python tools/ab.py --cfg headline --rounds 4 base=sprintz_amd/variants/base.so drop=sprintz_amd/variants/drop.so split=sprintz_amd/variants/split.so 2>&1 | grep -E "MEDIAN" | awk '{print $1,$2,$3,$4,$5}'
