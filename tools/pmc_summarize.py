#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc CSV (counter_collection.csv) per kernel: launches and summed counter value."""
import csv
import glob
import json
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?")
            c = row.get("Counter_Name", "?")
            a = agg[k][c]
            a[0] += 1
            v = float(row.get("Counter_Value", 0) or 0)
            a[1] += v
            a[2] = max(a[2], v)
res = {k: {c: {"dispatches": v[0], "sum": v[1], "max": v[2]} for c, v in d.items()} for k, d in agg.items()}
with open(out, "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
print("kernels:", len(res))
