#!/bin/bash
# Huff0 stage of the cfg4 chain at several batch sizes with the big-batch form (2-wave workgroups) forced and with the single-wave form forced:
# where SPRINTZ_OPT_HUF0_BIG_BATCH's default belongs.   tools/huf0_threshold_sweep.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for n in 5000 10000 20000 40000 80000; do
  for form in big small; do
    python - $n $form <<'PY'
import sys, json, subprocess, os
n, form = sys.argv[1], sys.argv[2]
code = f"""
import sys
sys.argv = ['bench.py', '--only', 'cfg4_{n}', '--no-cpu-baseline', '--config-reps', '20']
from sprintz_amd import _lib
_lib.set_option(_lib.OPT_HUF0_BIG_BATCH, {0 if form == 'big' else 10**9})
import runpy
runpy.run_path('bench.py', run_name='__main__')
"""
out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
print(n, form, d['huff0_decode_ms'], d['decompress_ms'])
PY
  done
done
