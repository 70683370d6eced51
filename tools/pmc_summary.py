#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per-dispatch averages of every counter for the run kernel."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "lbft_k_run"
prefix = sys.argv[3] if len(sys.argv) > 3 else ""  # only the pass directories whose name starts with this (one configuration)
acc = collections.defaultdict(list)
for path in glob.glob(root + "/" + prefix + "*/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel not in row.get("Kernel_Name", ""):
                continue
            per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (d, name), v in per_dispatch.items():
        acc[name].append(v)
print(json.dumps({k: sum(v) / len(v) for k, v in sorted(acc.items())} | {"_dispatches": {k: len(v) for k, v in acc.items()}}, indent=1))
