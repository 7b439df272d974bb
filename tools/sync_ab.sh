#!/bin/bash
# cfg4's small batches (the self-synchronising Huff0 stage) over library builds: tools/sync_ab.sh variants/a.so variants/b.so ...   (SIZES="625 1250 5000")
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for i in 1 2; do for n in ${SIZES:-1250 5000 8192}; do for L in "$@"; do echo -n "$L $n chunks: "; SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/$L timeout 300 python bench.py --only cfg4_$n --no-cpu-baseline --config-reps 30 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('huff0 us %.1f chain us %.1f' % (d['huff0_decode_ms']*1e3, d['decompress_ms']*1e3))"; done; done; done
