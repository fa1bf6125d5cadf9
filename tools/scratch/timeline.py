"""steady-state timeline of the pipelined bench from a rocprofv3 kernel trace: how the wall time between the first and the last
wg_hash_k launch of the timed steps splits into (a) idle, (b) only YD kernels running, (c) only coverage kernels, (d) window
kernels (+ anything).  usage: timeline.py kernel_trace.csv"""
import csv, sys, re, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"(\w+)_k\b", k)
    stem = m.group(1) if m else k[:30]
    if "so_two" in stem: stem = "yd_chains" if "SegMaxY" in k else "cov_bundles"
    if "w64_scatter" in stem and "YdEmit" in k: stem = "yd_scatter"
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), stem))
rows.sort()
wh = [r for r in rows if r[2] == "wg_hash"]
print("wg_hash launches", len(wh))
t0, t1 = wh[3][0], wh[-3][0]
nsteps = len(wh) - 6 + 1 - 1
def cat(s):
    if s.startswith("wg_") or s.startswith("col_") or s.startswith("g2c"): return "W"
    if s.startswith("yd_"): return "Y"
    if s.startswith("cov_") or s.startswith("junc_"): return "C"
    return "O"
ev = []
for b, e, s in rows:
    if e <= t0 or b >= t1: continue
    b, e = max(b, t0), min(e, t1)
    ev.append((b, 1, cat(s))); ev.append((e, -1, cat(s)))
ev.sort()
cnt = collections.Counter(); acc = collections.Counter(); last = t0
for t, d, c in ev:
    key = "".join(sorted(k for k in cnt if cnt[k] > 0)) or "idle"
    acc[key] += t - last; last = t
    cnt[c] += d
acc["idle"] += t1 - last
tot = t1 - t0
print("span %.2f ms over %d steps = %.2f ms/step" % (tot / 1e6, nsteps, tot / 1e6 / nsteps))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-6s %7.2f ms/step  %5.1f %%" % (k, v / 1e6 / nsteps, 100.0 * v / tot))
per = collections.Counter()
for b, e, s in rows:
    if b >= t0 and e <= t1: per[s] += e - b
print("kernel time/step:", ", ".join("%s %.2f" % (k, v / 1e6 / nsteps) for k, v in per.most_common(22)))
