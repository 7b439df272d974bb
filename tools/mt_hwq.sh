set -u
cd "$(dirname "$0")/.."
g++ -O2 -std=c++17 -o tools/probes/single_call_mt tools/probes/single_call_mt.cpp -ldl -lpthread
L=sprintz_amd/libsprintz_mi355x.so
for q in 8 16 32; do
echo "== GPU_MAX_HW_QUEUES=$q SPRINTZ_MI355X_HOST_STREAMS=$q"; GPU_MAX_HW_QUEUES=$q SPRINTZ_MI355X_HOST_STREAMS=$q timeout 120 tools/probes/single_call_mt $L 4 8 16 32 64 < /dev/null
done
echo "== GPU_MAX_HW_QUEUES=4 (default) SPRINTZ_MI355X_HOST_STREAMS=16"; SPRINTZ_MI355X_HOST_STREAMS=16 timeout 120 tools/probes/single_call_mt $L 8 16 32 64 < /dev/null
