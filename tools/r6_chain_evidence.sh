#!/bin/bash
# round 6's one-pass decoders against the forms they replace, and where a tile's time goes: tools/r6_chain_evidence.sh > gpurun_out/r6_chain.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
echo "== transforms decode, one pass (default) / two passes (SPRINTZ_MI355X_TRANSFORM_CHAIN=0): tools/transforms_ab.py 20"
timeout 200 python tools/transforms_ab.py 20 2>&1 | grep -v amdgpu.ids | tail -4
SPRINTZ_MI355X_TRANSFORM_CHAIN=0 timeout 200 python tools/transforms_ab.py 20 2>&1 | grep -v amdgpu.ids | tail -4 | sed 's/^default/twopass/'
echo "== dynamic delta, unpack in one pass (default) / three launches (SPRINTZ_MI355X_ONLINE_CHAIN=0): tools/online_ab.py 20 0"
timeout 200 python tools/online_ab.py 20 0 2>&1 | grep -v amdgpu.ids | tail -2
SPRINTZ_MI355X_ONLINE_CHAIN=0 timeout 200 python tools/online_ab.py 20 0 2>&1 | grep -v amdgpu.ids | tail -2 | sed 's/^dynamic_delta /three-launch   /'
if [ -f sprintz_amd/variants/chain_ts.so ]; then
  echo "== a tile's phases, -DTR_CHAIN_TIMING build (tools/chain_phases.py): delta, double delta"
  SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/chain_ts.so timeout 100 python tools/chain_phases.py delta 2>&1 | grep -v amdgpu.ids
  SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/variants/chain_ts.so timeout 100 python tools/chain_phases.py doubledelta 2>&1 | grep -v amdgpu.ids
fi
echo "== SQ counters of chain_scan_kernel (delta, 64 Mi samples): tools/pmc_cmd.sh"
tools/pmc_cmd.sh pmc_chain_final chain_scan python $PWD/tools/transforms_ab.py 5 delta 64Mi 2>/dev/null | cut -c60-
