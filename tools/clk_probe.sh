#!/bin/bash
# clocks, socket power and temperature WHILE the headline decode loops (a 60 000-launch run, ~25 s): is the part at its power cap?
#   tools/clk_probe.sh > gpurun_out/clk_probe.txt      (sample lines: every 2 s from second 6 on)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python bench.py --configs none --no-extras --no-cpu-baseline --steps 60000 --warmup 3 > /tmp/clk_b.json 2>/tmp/clk_b.err &
BP=$!
sleep 6
n=0
while kill -0 $BP 2>/dev/null && [ $n -lt 12 ]; do
  echo "--- sample $n (headline decode looping)"
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | head -8
  n=$((n+1)); sleep 2
done
wait $BP
echo "--- idle, 3 s after the loop"
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
python -c "
import json; d=json.load(open('/tmp/clk_b.json')); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'])"
