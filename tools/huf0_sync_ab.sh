#!/bin/bash
# A/B of the Huff0 stream stage for small batches: huf0_sync.h (a wave per chunk, sixteen self-synchronising decoders per stream) against
# the single-wave form it replaces there, over batch sizes (tools/huf0_sync_ab.sh > gpurun_out/huf0_sync_ab.txt)
# SYNCS: values of SPRINTZ_MI355X_HUF0_SYNC_CHUNKS to compare; CPWS: (round 5 only: the knob was removed in round 6, one chunk a wave)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for sync in ${SYNCS:-0 1073741824}; do
 for cpw in ${CPWS:-0}; do
  echo "SPRINTZ_MI355X_HUF0_SYNC_CHUNKS=$sync SPRINTZ_MI355X_HUF0_SYNC_CPW=$cpw"
  echo "| chunks | chain us | Huff0 us | Sprintz us |"
  for n in ${SIZES:-625 1250 2500 5000 10000 20000 40000 80000}; do
    SPRINTZ_MI355X_HUF0_SYNC_CPW=$cpw SPRINTZ_MI355X_HUF0_SYNC_CHUNKS=$sync timeout 300 python bench.py --only cfg4_$n --no-cpu-baseline --config-reps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('| %d | %.1f | %.1f | %.1f |' % (d['chunks'], d['decompress_ms'] * 1e3, d['huff0_decode_ms'] * 1e3, d['sprintz_decode_ms'] * 1e3))"
  done
 done
done
