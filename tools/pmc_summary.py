#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch.
Usage: pmc_summary.py <counter_collection.csv> [out.csv] [name filter substring]"""
import csv, sys, collections
src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
flt = sys.argv[3] if len(sys.argv) > 3 else None
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
with open(src) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
rows = [["kernel", "dispatches"] + names]
for k in sorted(acc):
    n = max(v[1] for v in acc[k].values())
    rows.append([k[:100], n] + ["%.1f" % (acc[k][c][0] / max(acc[k][c][1], 1)) if c in acc[k] else "" for c in names])
w = csv.writer(open(out, "w") if out else sys.stdout)
w.writerows(rows)
