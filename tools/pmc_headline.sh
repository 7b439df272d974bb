#!/bin/bash
# SQ counters of the headline decode kernel: tools/pmc_headline.sh <outdir-name> [lib.so]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; LIB=$2
mkdir -p $OUT
[ -n "$LIB" ] && export SPRINTZ_MI355X_LIB=$PWD/$LIB
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --configs none --no-extras --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 > /dev/null 2> $OUT/pmc_$i.err)
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH
GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_EXP_GDS
GROUPS
python tools/pmc_report.py $OUT decode_fast | awk '{print $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
