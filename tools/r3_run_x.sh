cd $GRAFT_REPO_ROOT
echo "== tests"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/base.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_colmajor.py tests/test_gpu_bench_data.py -m gpu -x -q 2>&1 | tail -2
python tools/ab.py --cfg headline --rounds 4 prev=sprintz_amd/variants/prev.so lines=sprintz_amd/variants/base.so 2>&1 | grep -E "MEDIAN"
