#!/bin/bash
# interleaved A/B of two library builds on the same box: tools/ab.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    echo -n "$(basename $L) "
    SPRINTZ_MI355X_LIB=$PWD/$L timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['roofline']['frac'], 'compress', d['compress_MBps'])"
  done
done
