#!/usr/bin/env python3
"""prints the kernel and memory-copy summaries of a rocprofv3 --stats run (the csv files under the given directory)"""
import csv, glob, os, sys
d = sys.argv[1]
for pat in ("*kernel_stats.csv", "*memory_copy_stats.csv"):
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        print("==", f)
        rows = list(csv.DictReader(open(f)))
        for r in rows[:25]:
            print("  %-60s calls %6s total %10.3f ms avg %10.3f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    big = sorted(rows, key=lambda r: -(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))[:12]
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    print("== largest copies of", f)
    for r in big:
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        print("  %-28s start %8.1f ms dur %8.2f ms" % (r.get("Direction", r.get("Name", "?")), (int(r["Start_Timestamp"]) - t0) / 1e6, dt), {k: r[k] for k in r if "ytes" in k or "ize" in k})
