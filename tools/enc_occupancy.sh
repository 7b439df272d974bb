# what occupancy is worth to the headline encoder: unused LDS claimed per workgroup -> 4 / 3 / 2 / 1 workgroups (16 / 12 / 8 / 4 waves) a CU
cd "$(dirname "$0")/.."
for pad in 0 2000 15000 42000 100000; do echo -n "pad $pad: "; SPRINTZ_MI355X_ENC_LDS_PAD=$pad timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --configs none --no-extras --no-sweep --no-verify 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress ms', d['compress']['ms_per_step_max_rank'], d['compress']['two_launch_ms_this_rank'])"; done
