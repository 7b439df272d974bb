cd $GRAFT_REPO_ROOT
for v in lean leanhdr always encpk; do
  echo "== tests with $v"; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/$v.so python -m pytest tests -m gpu -x -q 2>&1 | tail -3
done
echo "== new tests (default lib)"; python -m pytest tests/test_gpu_bench_data.py -m gpu -x -q 2>&1 | tail -5
python tools/ab.py --cfg headline --rounds 3 base=sprintz_amd/variants/base.so lean=sprintz_amd/variants/lean.so leanhdr=sprintz_amd/variants/leanhdr.so always=sprintz_amd/variants/always.so encpk=sprintz_amd/variants/encpk.so 2>&1 | tail -22
