cd $GRAFT_REPO_ROOT
python bench.py --configs none --no-extras --no-cpu-baseline --steps 20000 --warmup 3 > /tmp/b.json 2>/tmp/b.err &
BP=$!
sleep 12
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | head -8; echo ---; sleep 1.5; done
wait $BP
python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['ms_per_step'], d['kernel_ms'])"
