#!/bin/bash
# SQ / TCP counters of the big-batch Huff0 stream kernel (huf0_stream_kernel<true, 2, 6, ...>) at cfg4's 800 000 chunks: where a wave's cycles go.
# One rocprofv3 pass per group, --kernel-trace only.      tools/pmc_huf0_big.sh <outdir under gpurun_out> [chunks]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$1
N=${2:-800000}
mkdir -p $OUT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $ROOT/bench.py --only cfg4_$N --no-cpu-baseline --config-reps 3 --no-verify > $OUT/p$i.json 2> $OUT/p$i.err < /dev/null)
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_LEVEL_WAVES SQ_WAVES SQ_THREAD_CYCLES_VALU
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC
GROUPS
python tools/pmc_report.py $OUT huf0_stream | tee $OUT/report.txt
find $OUT -name "*.csv" -size +1000k -delete; rm -rf $OUT/p*/*/*.db 2>/dev/null
