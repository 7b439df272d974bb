#!/bin/bash
# durations of, and gaps between, consecutive kernels of a chain:   tools/kgaps.sh <outdir-name> <name-substring,name-substring,...> <command...>
#   e.g. tools/kgaps.sh kg huf0_tree_wave,huf0_stream_small,decode_fast python /root/repo/bench.py --only cfg4_10000 --no-cpu-baseline --config-reps 20
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$1; KEYS=$2; shift 2
mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o r -- "$@" > $OUT/cmd.out 2> $OUT/cmd.err < /dev/null)
python - <<PY
import glob, sqlite3, statistics
f = glob.glob("$OUT/trace/**/*.db", recursive=True)
if not f:
    raise SystemExit("no rocprofv3 database under $OUT/trace")
db = sqlite3.connect(f[0])
rows = list(db.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
keys = "$KEYS".split(",")
hits = [i for i in range(len(rows) - len(keys) + 1) if all(k in rows[i + j][2] for j, k in enumerate(keys))]
med = statistics.median
out = []
for j, k in enumerate(keys):
    out.append(f"{k} {med([rows[i + j][1] - rows[i + j][0] for i in hits]) / 1e3:.1f} us")
    if j + 1 < len(keys):
        out.append(f"gap {med([rows[i + j + 1][0] - rows[i + j][1] for i in hits]) / 1e3:.1f} us")
print(f"{len(hits)} chains:  " + "  |  ".join(out))
print(f"first start to last end: {med([rows[i + len(keys) - 1][1] - rows[i][0] for i in hits]) / 1e3:.1f} us")
PY
rm -rf $OUT/trace
