cd /root/repo
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/ta_all; mkdir -p $OUT
(cd /tmp && timeout 400 rocprofv3 --pmc TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ta -o p -- python /root/repo/bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-sweep --config-reps 3 > /dev/null 2> $OUT/err.txt < /dev/null)
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/ta_all/ta/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
rows=[]
for k,v in acc.items():
    if 'TA_BUSY_avr' in v and 'GRBM_GUI_ACTIVE' in v:
        ta=sum(v['TA_BUSY_avr'])/len(v['TA_BUSY_avr']); gui=sum(v['GRBM_GUI_ACTIVE'])/len(v['GRBM_GUI_ACTIVE'])/8
        if gui>20000: rows.append((ta/gui,ta,gui,len(v['TA_BUSY_avr']),k))
for r in sorted(rows,reverse=True)[:40]: print("%.2f ta %9.0f cyc %9.0f n=%3d %s"%r)
PY
