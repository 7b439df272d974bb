cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for m in 0 1 2; do echo -n "dense mode $m: "; SPRINTZ_MI355X_DENSE_MODE=$m python bench.py --configs none --no-extras --no-cpu-baseline --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['compress']['ms_per_step_max_rank'], d['compress']['two_launch_ms_this_rank'], d['kernel_ms'])"; done
for m in 0 2; do for c in cfg3_10k cfg4_800000; do echo -n "mode $m $c: "; SPRINTZ_MI355X_DENSE_MODE=$m python bench.py --only $c --no-cpu-baseline --config-reps 8 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['compress_ms'], d.get('huff0_encode_ms'))"; done; done
