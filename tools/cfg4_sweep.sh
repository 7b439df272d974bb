#!/bin/bash
# BASELINE config 4's decode chain (Huff0 wire format -> Sprintz) over batch sizes: us per batch, TB/s of samples -- where the chain turns
# from three serial latencies into bandwidth (tools/cfg4_sweep.sh > gpurun_out/cfg4_sweep.txt)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
echo "| chunks | chain us | Huff0 us | Sprintz us | TB/s of samples |"
echo "|---|---|---|---|---|"
for n in 625 1250 2500 5000 10000 20000 40000 80000 160000 800000; do
  timeout 200 python bench.py --only cfg4_$n --no-cpu-baseline --config-reps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('| %d | %.1f | %.1f | %.1f | %.3f |' % (d['chunks'], d['decompress_ms'] * 1e3, d['huff0_decode_ms'] * 1e3, d['sprintz_decode_ms'] * 1e3, d['raw_bytes'] / d['decompress_ms'] / 1e9))"
done
