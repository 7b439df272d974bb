#!/bin/bash
# encode rates of narrow row-major shapes with one column per lane (encode_fast.h) and two (encode_wide.h, DPT): tools/enc_pair_sweep.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for shape in "xff 2 8 5120" "xff 1 8 8192" "delta 2 16 5120" "xff 2 32 5120" "xff 2 64 5120" "delta 1 16 10240" "xff 1 32 10240" "xff 1 64 10240" "xff 2 6 3840" "delta 2 12 3840"; do
  for p in 0 1; do
    echo -n "pair=$p  "; SPRINTZ_MI355X_ENC_PAIR=$p python tools/bench_shape.py $shape 256 2>&1 | tail -1
  done
done
