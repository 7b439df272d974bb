#!/usr/bin/env python3
"""Aggregate the per-pass counter_collection.csv files of tools/pmc_sweep.sh:
average value per dispatch of every counter, for kernels matching a substring."""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "decode")
acc = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:40s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
