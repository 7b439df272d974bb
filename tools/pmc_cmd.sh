#!/bin/bash
# SQ counters of one kernel under ANY command: tools/pmc_cmd.sh <outdir under gpurun_out> <kernel substring> <command...>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$1; PAT=$2; shift 2
mkdir -p $OUT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p -- "$@" > $OUT/p$i.json 2> $OUT/p$i.err < /dev/null)
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_IFETCH SQ_LEVEL_WAVES SQ_WAVES SQ_THREAD_CYCLES_VALU
GROUPS
python tools/pmc_report.py $OUT "$PAT" | tee $OUT/report.txt
find $OUT -name "*.csv" -size +1000k -delete; rm -rf $OUT/p*/*/*.db 2>/dev/null
