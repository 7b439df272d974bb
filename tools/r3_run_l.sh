cd $GRAFT_REPO_ROOT
for v in sp6 sp6w4 sp5 sp5w4; do echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests/test_gpu_huf0.py -m gpu -x -q 2>&1 | tail -2; done
python tools/ab.py --cfg cfg4_800000 --rounds 2 --reps 8 spec=sprintz_amd/variants/spec.so sp5=sprintz_amd/variants/sp5.so sp5w4=sprintz_amd/variants/sp5w4.so sp6=sprintz_amd/variants/sp6.so sp6w4=sprintz_amd/variants/sp6w4.so 2>&1 | tail -16
