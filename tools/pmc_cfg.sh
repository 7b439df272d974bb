#!/bin/bash
# SQ counters of one bench configuration's kernels: tools/pmc_cfg.sh <outdir-name> <cfg> <kernel-substr> [lib.so]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; CFG=$2; PAT=$3; LIB=$4
mkdir -p $OUT
[ -n "$LIB" ] && export SPRINTZ_MI355X_LIB=$PWD/$LIB
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/pmc_${CFG}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --only $CFG --no-cpu-baseline --config-reps 3 > /dev/null 2> $OUT/pmc_${CFG}_$i.err)
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH
GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC
GROUPS
python tools/pmc_report.py $OUT $PAT | grep -v "^void at::\|rocprim"
