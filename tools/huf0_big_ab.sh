#!/bin/bash
# the Huff0 stage of BASELINE config 4 at 800 000 chunks over builds of the big-batch stream kernel (ring slots, waves a workgroup, piece size):
#   tools/huf0_big_ab.sh variants/base.so variants/h_ns2.so ...      (paths under sprintz_amd/; two alternating passes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for i in 1 2; do for L in "$@"; do echo -n "$L "; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/$L timeout 300 python bench.py --only cfg4_${CHUNKS:-800000} --no-cpu-baseline --config-reps 8 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('huff0 ms', d['huff0_decode_ms'], 'chain ms', d['decompress_ms'], 'frac', d['roofline']['frac'])"; done; done
