cd $GRAFT_REPO_ROOT
for v in l128w4 l128w5; do echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests/test_gpu_huf0.py tests/test_gpu_bench_data.py -m gpu -x -q 2>&1 | tail -2; done
python tools/ab.py --cfg cfg4_800000 --rounds 2 --reps 8 l64=sprintz_amd/variants/l64.so l128w5=sprintz_amd/variants/l128w5.so l128w4=sprintz_amd/variants/l128w4.so 2>&1 | grep -E "round" | awk '{print $3, $5, $9}'
python tools/ab.py --cfg cfg4_10000 --rounds 2 --reps 20 l64=sprintz_amd/variants/l64.so l128w5=sprintz_amd/variants/l128w5.so l128w4=sprintz_amd/variants/l128w4.so 2>&1 | grep -E "round" | awk '{print $3, $5, $9}'
