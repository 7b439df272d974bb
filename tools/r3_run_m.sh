cd $GRAFT_REPO_ROOT
for v in sp5w2 sp4w4 sp5w4e6; do echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests/test_gpu_huf0.py -m gpu -x -q 2>&1 | tail -2; done
python tools/ab.py --cfg cfg4_800000 --rounds 2 --reps 8 spec=sprintz_amd/variants/spec.so sp5w2=sprintz_amd/variants/sp5w2.so sp5w4=sprintz_amd/variants/sp5w4.so sp4w4=sprintz_amd/variants/sp4w4.so sp5w4e6=sprintz_amd/variants/sp5w4e6.so 2>&1 | tail -7
python tools/ab.py --cfg cfg4_80000 --rounds 2 --reps 10 spec=sprintz_amd/variants/spec.so sp5w4=sprintz_amd/variants/sp5w4.so sp5w4e6=sprintz_amd/variants/sp5w4e6.so 2>&1 | tail -4
python tools/ab.py --cfg cfg4_10000 --rounds 2 --reps 20 spec=sprintz_amd/variants/spec.so sp5w4=sprintz_amd/variants/sp5w4.so sp5w4e6=sprintz_amd/variants/sp5w4e6.so 2>&1 | tail -4
