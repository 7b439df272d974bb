#!/usr/bin/env python3
"""Turn the rocprofv3 databases of tools/profile_round2.sh (gpurun_out/<run>/<workload>/{trace,pmc_fetch,pmc_write})
into the text summaries committed under profiles/:

    python profiles/summarize_r2.py gpurun_out/prof_r2 profiles/r2 --build "$(git rev-parse --short HEAD)"

  <prefix>_<workload>_kernel_stats.csv   the `rocprofv3 --kernel-trace --stats` table of that workload's command
  <prefix>_<workload>_pmc.csv            FETCH_SIZE / WRITE_SIZE per kernel (averages per dispatch)
  profiles/hbm_traffic.json              {"entries": {kernel substring: bytes per launch, source, build, ...}} -- what bench.py
                                         quotes as roofline.traffic, labelled with the build it was measured on

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are KiB, collected in separate --pmc passes; on
gfx950 FETCH_SIZE reports half of a wide coalesced read stream (128-byte requests tallied at 64 B), so it is doubled for
the kernels whose reads are 16 bytes per lane and contiguous (decode_fast, compact_copy); WRITE_SIZE is used as is."""
import argparse
import csv
import glob
import json
import os
import sqlite3


def one_db(d):
    files = glob.glob(os.path.join(d, "*.db"))
    return sqlite3.connect(files[0]) if files else None


OURS = ("sprintz", "huf", "compact_copy", "scan_", "zigzag_kernel", "dyndelta", "pack_w", "unpack_", "xff_kernel", "verbatim_")

# kernel substring quoted by bench.py -> (workload whose PMC passes hold it, nchunks, data)
DOMINANT = {"decode_fast_kernel<16, true, 8": ("headline", 131072, "walk8")}
# workloads whose decode is a CHAIN of kernels: the chain's HBM bytes = the sum over its kernels of the average bytes per dispatch
# (every distinct kernel runs once per chain; FETCH_SIZE doubled for all of them: their reads are 16 bytes a lane and contiguous)
CHAIN_KERNELS = ("huf0_follow", "huf0_copy", "huf0_tree", "huf0_share", "huf0_stream", "huf0_sync", "decode_fast_kernel", "decode_lat_kernel")
CHAINS = ("cfg4_1250", "cfg4_10000", "cfg4_80000", "cfg4_800000")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("run_dir")
    ap.add_argument("out_prefix")
    ap.add_argument("--build", default="unlabelled")
    ap.add_argument("--traffic-json", default=None, help="where to write hbm_traffic.json (default: next to out_prefix)")
    a = ap.parse_args()
    entries = {}
    for wdir in sorted(glob.glob(os.path.join(a.run_dir, "*"))):
        w = os.path.basename(wdir)
        if not os.path.isdir(wdir) or w == "summary":
            continue
        db = one_db(os.path.join(wdir, "trace"))
        if db:
            rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
            with open(f"{a.out_prefix}_{w}_kernel_stats.csv", "w", newline="") as f:
                wr = csv.writer(f)
                wr.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
                for r in rows:
                    if any(k in r[0] for k in OURS):
                        wr.writerow([r[0][:160], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.2f}"])
        pmc = {}
        for sub in ("pmc_fetch", "pmc_write"):
            db = one_db(os.path.join(wdir, sub))
            if not db:
                continue
            q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                 "from counters_collection group by kernel_name, counter_name")
            try:
                for k, c, n, avg, lo, hi, dur in db.execute(q):
                    pmc.setdefault(k, {})[c] = (n, avg, lo, hi, dur)
            except sqlite3.Error:
                pass
        if pmc:
            with open(f"{a.out_prefix}_{w}_pmc.csv", "w", newline="") as f:
                wr = csv.writer(f)
                wr.writerow(["Kernel", "Counter", "Dispatches", "AvgKiB", "MinKiB", "MaxKiB", "AvgDurationNs"])
                for k in sorted(pmc):
                    if not any(s in k for s in OURS):
                        continue
                    for c, v in sorted(pmc[k].items()):
                        wr.writerow([k[:120], c, v[0], f"{v[1]:.2f}", f"{v[2]:.2f}", f"{v[3]:.2f}", f"{v[4]:.0f}"])
        for sub, (wl, nchunks, data) in DOMINANT.items():
            if wl != w:
                continue
            dom = [k for k in pmc if sub in k]
            if dom:
                k = dom[0]
                fetch = pmc[k].get("FETCH_SIZE", (0, 0))[1] * 1024 * 2      # gfx950 correction (wide coalesced reads)
                write = pmc[k].get("WRITE_SIZE", (0, 0))[1] * 1024
                entries[sub] = dict(kernel=k, nchunks=nchunks, data=data, fetch_bytes_corrected=int(fetch), write_bytes=int(write),
                                    bytes_per_launch=int(fetch + write), source=os.path.basename(a.out_prefix) + f"_{w}_pmc.csv", build=a.build,
                                    method="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), KiB->B")
        if w in CHAINS and pmc:
            tot_f = tot_w = 0.0
            parts = {}
            for k in pmc:
                if not any(c in k for c in CHAIN_KERNELS):
                    continue
                f = pmc[k].get("FETCH_SIZE", (0, 0))[1] * 1024 * 2
                wv = pmc[k].get("WRITE_SIZE", (0, 0))[1] * 1024
                tot_f += f
                tot_w += wv
                parts[k[:70]] = int(f + wv)
            if parts:
                entries["chain:" + w] = dict(kernel="the decode chain of " + w, fetch_bytes_corrected=int(tot_f), write_bytes=int(tot_w),
                                             bytes_per_launch=int(tot_f + tot_w), parts=parts, source=os.path.basename(a.out_prefix) + f"_{w}_pmc.csv",
                                             build=a.build, method="sum over the chain's kernels of (FETCH_SIZE x2 + WRITE_SIZE) per dispatch, separate --pmc passes")
    if entries:
        with open(a.traffic_json or os.path.join(os.path.dirname(a.out_prefix) or ".", "hbm_traffic.json"), "w") as f:
            json.dump({"entries": entries}, f, indent=1)
    print(json.dumps(entries, indent=1))


if __name__ == "__main__":
    main()
