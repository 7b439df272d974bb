cd $GRAFT_REPO_ROOT
echo "== tests with merge8"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/merge8.so python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/ab.py --cfg cfg3_10k --rounds 3 base=sprintz_amd/variants/base.so merge8=sprintz_amd/variants/merge8.so 2>&1 | tail -8
