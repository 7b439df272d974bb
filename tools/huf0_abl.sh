#!/bin/bash
# ablation builds of the big-batch Huff0 stream kernel (no verification: they decode garbage on purpose): tools/huf0_abl.sh variants/a.so ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for i in 1 2; do for L in "$@"; do echo -n "$L "; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/$L timeout 300 python bench.py --only cfg4_${CHUNKS:-800000} --no-cpu-baseline --no-verify --config-reps 8 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('huff0 ms', d['huff0_decode_ms'], 'chain ms', d['decompress_ms'])"; done; done
