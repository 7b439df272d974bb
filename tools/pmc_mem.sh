#!/bin/bash
# memory-side counters of one bench configuration: tools/pmc_mem.sh <outdir-name> <cfg> <kernel-substr> [lib.so]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; CFG=$2; PAT=$3; LIB=$4
mkdir -p $OUT
[ -n "$LIB" ] && export SPRINTZ_MI355X_LIB=$PWD/$LIB
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/mem_${CFG}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --only $CFG --no-cpu-baseline --config-reps 3 > /dev/null 2> $OUT/mem_${CFG}_$i.err)
done <<'GROUPS'
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUSY_max
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
GROUPS
python tools/pmc_report.py $OUT $PAT | awk '{print $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
