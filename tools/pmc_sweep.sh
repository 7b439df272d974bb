#!/bin/bash
# PMC sweep of bench.py's dominant kernel: one rocprofv3 pass per counter group
# (counters in their own runs, --kernel-trace only, as the gpurun policy requires).
# usage: tools/pmc_sweep.sh <outdir> [bench args...]
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/p$i.json 2> $OUT/p$i.err
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_LEVEL_WAVES
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
GROUPS
find $OUT -name "*counter_collection.csv" | head -20
