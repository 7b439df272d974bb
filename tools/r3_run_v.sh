cd $GRAFT_REPO_ROOT
for v in t128 t512; do echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2; done
python tools/ab.py --cfg headline --rounds 3 base=sprintz_amd/variants/base.so t128=sprintz_amd/variants/t128.so t512=sprintz_amd/variants/t512.so 2>&1 | grep -E "MEDIAN"
