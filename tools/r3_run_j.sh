cd $GRAFT_REPO_ROOT
echo "== tests with batch"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/batch.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_data.py tests/test_gpu_norle.py tests/test_gpu_query.py tests/test_gpu_colmajor.py -m gpu -x -q 2>&1 | tail -3
python tools/ab.py --cfg headline --rounds 4 base=sprintz_amd/variants/base.so batch=sprintz_amd/variants/batch.so 2>&1 | tail -12
