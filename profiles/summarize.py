#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (gpurun_out/<run>/{trace,pmc_fetch,pmc_write}/*_results.db)
into the small text summaries committed under profiles/.

    python profiles/summarize.py gpurun_out/prof_r1 profiles/r1_v1_generic --nchunks 131072 --data walk8

Writes <out>_kernel_stats.csv (the `rocprofv3 --kernel-trace --stats` table), <out>_pmc.csv
(per-kernel FETCH_SIZE / WRITE_SIZE averages) and, for the decode kernel, updates
profiles/hbm_traffic.json which bench.py reports as roofline.traffic.

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream
(128-byte requests tallied at 64 B) so it is doubled; WRITE_SIZE is used as is.  Both corrections
were re-checked in the same run against torch elementwise kernels of known size (int32 AND over
1.342 GB per launch: FETCH_SIZE 655371 KiB = 0.5x, WRITE_SIZE 1310720 KiB = 1.0x).
"""
import argparse
import csv
import glob
import json
import os
import sqlite3


def one_db(d):
    files = glob.glob(os.path.join(d, "*.db"))
    return sqlite3.connect(files[0]) if files else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("run_dir")
    ap.add_argument("out_prefix")
    ap.add_argument("--nchunks", type=int, default=131072)
    ap.add_argument("--data", default="walk8")
    ap.add_argument("--kernel", default="decode_kernel", help="substring of the dominant kernel's name")
    a = ap.parse_args()

    db = one_db(os.path.join(a.run_dir, "trace"))
    if db:
        rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        with open(a.out_prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            for r in rows:
                w.writerow([r[0][:160], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.2f}"])
    pmc = {}
    for sub in ("pmc_fetch", "pmc_write"):
        db = one_db(os.path.join(a.run_dir, sub))
        if not db:
            continue
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
             "from counters_collection group by kernel_name, counter_name")
        for k, c, n, avg, lo, hi, dur in db.execute(q):
            pmc.setdefault(k, {})[c] = (n, avg, lo, hi, dur)
    with open(a.out_prefix + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "AvgKiB", "MinKiB", "MaxKiB", "AvgDurationNs"])
        for k in sorted(pmc):
            if "sprintz" not in k and "compact" not in k and "scan_" not in k:
                continue
            for c, v in sorted(pmc[k].items()):
                w.writerow([k[:120], c, v[0], f"{v[1]:.2f}", f"{v[2]:.2f}", f"{v[3]:.2f}", f"{v[4]:.0f}"])
    dom = [k for k in pmc if a.kernel in k]
    if dom:
        k = dom[0]
        fetch = pmc[k].get("FETCH_SIZE", (0, 0))[1] * 1024 * 2      # gfx950 correction
        write = pmc[k].get("WRITE_SIZE", (0, 0))[1] * 1024
        out = dict(kernel=k, nchunks=a.nchunks, data=a.data, fetch_bytes_corrected=int(fetch), write_bytes=int(write),
                   bytes_per_launch=int(fetch + write), source=os.path.basename(a.out_prefix) + "_pmc.csv",
                   method="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), KiB->B")
        with open(os.path.join(os.path.dirname(a.out_prefix) or ".", "hbm_traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
