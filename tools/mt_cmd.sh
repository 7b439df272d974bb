# thread scaling of the drop-in single calls from native threads:   bash tools/mt_cmd.sh     (on the GPU box)
set -u
cd "$(dirname "$0")/.."
g++ -O2 -std=c++17 -o tools/probes/single_call_mt tools/probes/single_call_mt.cpp -ldl -lpthread
L=sprintz_amd/libsprintz_mi355x.so
for w in 0 1 2; do
echo "== SPRINTZ_MI355X_HOST_WAIT=$w"; SPRINTZ_MI355X_HOST_WAIT=$w timeout 120 tools/probes/single_call_mt $L 1 2 4 6 8 12 16 32 64 128 < /dev/null
done
echo "== SPRINTZ_MI355X_HOST_STREAMS=8"; SPRINTZ_MI355X_HOST_STREAMS=8 timeout 120 tools/probes/single_call_mt $L 8 12 16 32 64 < /dev/null
echo "== cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
