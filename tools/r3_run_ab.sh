cd $GRAFT_REPO_ROOT
echo "== tests with wf1"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/base.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_data.py tests/test_gpu_query.py -m gpu -x -q 2>&1 | tail -2
python tools/ab.py --cfg cfg1 --rounds 4 wf0=sprintz_amd/variants/wf0.so wf1=sprintz_amd/variants/base.so 2>&1 | grep -E "MEDIAN"
