cd $GRAFT_REPO_ROOT
for v in enc_new16 enc_d128 enc_d64; do echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_colmajor.py tests/test_gpu_norle.py tests/test_gpu_bench_data.py -m gpu -x -q 2>&1 | tail -2; done
python tools/ab.py --cfg headline --rounds 3 --no-verify base=sprintz_amd/variants/base.so enc_nostore=sprintz_amd/variants/enc_nostore.so enc_d64=sprintz_amd/variants/enc_d64.so enc_d128=sprintz_amd/variants/enc_d128.so 2>&1 | grep -E "MEDIAN"
