#!/bin/bash
# the Sprintz stage of small cfg4 batches over builds of decode_lat: tools/lat_ab.sh variants/base.so variants/sl4.so ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for n in ${SIZES:-625 1250 2048}; do for L in "$@"; do echo -n "$n $L "; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/$L timeout 200 python bench.py --only cfg4_$n --no-cpu-baseline --config-reps 20 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sprintz us', round(d['sprintz_decode_ms']*1e3,1), 'chain us', round(d['decompress_ms']*1e3,1))"; done; done
