cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in fused twolaunch; do
  if [ $mode = twolaunch ]; then export SPRINTZ_MI355X_NO_FUSED_COMPACT=1; else unset SPRINTZ_MI355X_NO_FUSED_COMPACT; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof_enc_$mode -o p -- python $GRAFT_REPO_ROOT/bench.py --configs none --no-extras --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3_prof_enc_$mode.err)
  f=$(find gpurun_out/r3_prof_enc_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; head -12 $f | cut -c1-200
done
