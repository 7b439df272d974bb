cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in base ablstore; do   # tools/build_variant.sh base ""; tools/build_variant.sh ablstore "decode_w16" -DSPRINTZ_ABL_NO_GSTORE
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  (cd /tmp && SPRINTZ_MI355X_LIB=$GRAFT_REPO_ROOT/sprintz_amd/variants/$v.so timeout 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_pmcst_${v}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --configs none --no-extras --no-cpu-baseline --no-verify --steps 3 --warmup 1 --ramp-ms 0 > /dev/null 2>/dev/null)
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_max TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
GROUPS
echo "== $v"; python tools/pmc_report.py gpurun_out decode_fast 2>/dev/null | grep "r3_pmcst_${v}_" > /dev/null; python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r3_pmcst_${v}_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "decode_fast" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,vv in sorted(acc.items()): print("  %-40s %16.0f" % (k, sum(vv)/len(vv)))
PY
done
