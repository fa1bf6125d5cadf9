#!/usr/bin/env python3
"""Host-clock time of each call of one bench step (every C-ABI call returns synchronised, so these add up to the step):
collapse (main stage; the YD stage is left running on its side context), groups_to_cov_in, coverage, finish_yd (what
is still left of the YD stage).  Usage: stage_times.py [profile files reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiebrush_amd import api, synth

prof = sys.argv[1] if len(sys.argv) > 1 else "c2"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
kw = {"c2": {}, "c3": dict(strategy="clip"), "c5": dict(strategy="exon", max_nh=5, min_qual=1)}[prof]
tile = synth.make_tile(nf, nr, prof)
ctx = api.Context(0)
dt = api.to_device(tile, "cuda:0")
acc = [0.0] * 5
N = 40
for it in range(N + 5):
    t0 = time.perf_counter()
    g = ctx.collapse(dt, defer_yd=True, want_coords=True, **kw)
    t1 = time.perf_counter()
    v = ctx.groups_to_cov_in(g)
    t2 = time.perf_counter()
    c = ctx.coverage(v)
    t3 = time.perf_counter()
    ctx.finish_yd()
    t4 = time.perf_counter()
    if it >= 5:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
            acc[i] += d
print("ms/step: collapse %.3f  groups_to_cov_in %.3f  coverage %.3f  finish_yd(wait) %.3f  total %.3f" % tuple(1e3 * a / N for a in acc))
# the YD stage alone (not deferred)
t0 = time.perf_counter()
for it in range(N):
    g = ctx.collapse(dt, want_coords=True, **kw)
print("collapse with inline YD: %.3f ms" % (1e3 * (time.perf_counter() - t0) / N))
t0 = time.perf_counter()
for it in range(N):
    c = ctx.coverage(v)
print("coverage alone: %.3f ms" % (1e3 * (time.perf_counter() - t0) / N))
