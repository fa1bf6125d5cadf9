#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: launches, summed kernel time and GPU-busy time (union of the kernel intervals) per
bench step, and the top kernels by total time.  Usage: trace_busy.py <kernel_trace.csv> <steps incl. warm-up> [skip substring ...]
Kernels whose name contains a skip substring (e.g. the synthetic generator's torch kernels) are left out."""
import csv, sys, collections
src, steps = sys.argv[1], int(sys.argv[2])
skip = sys.argv[3:]
iv, per = [], collections.defaultdict(lambda: [0, 0])
with open(src) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if any(s in k for s in skip):
            continue
        b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        iv.append((b, e))
        per[k][0] += 1
        per[k][1] += e - b
iv.sort()
busy, cb, ce = 0, None, None
for b, e in iv:
    if cb is None:
        cb, ce = b, e
    elif b <= ce:
        ce = max(ce, e)
    else:
        busy += ce - cb
        cb, ce = b, e
if cb is not None:
    busy += ce - cb
tot = sum(e - b for b, e in iv)
print("launches/step %.1f | summed kernel time %.3f ms/step | GPU busy (union) %.3f ms/step | span %.3f ms/step" %
      (len(iv) / steps, tot / steps / 1e6, busy / steps / 1e6, (iv[-1][1] - iv[0][0]) / steps / 1e6))
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%8.1f us/step %6.2f calls/step %8.1f us avg  %s" % (t / steps / 1e3, n / steps, t / n / 1e3, k[:90]))
