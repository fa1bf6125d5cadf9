#!/usr/bin/env python3
"""Per-kernel times of the tiecov stage on the collapsed records of a synthetic tile (default: config 3 at full size).
Usage: cov_prof.py [profile files reads reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiebrush_amd import api, synth_dev

prof = sys.argv[1] if len(sys.argv) > 1 else "c3"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kw = {"c2": {}, "c3": dict(strategy="clip"), "c5": dict(strategy="exon", max_nh=5, min_qual=1)}[prof]
sp = {"c2": "c2", "c3": "c3", "c5": "c5"}[prof]
dt = synth_dev.make_tile_device(nf, nr, sp, device="cuda:0")
ctx = api.Context(0)
g = ctx.collapse(dt, **kw)
view = ctx.groups_to_cov_in(g)
bufs = {}
WJ = os.environ.get("TBK_PROF_NOJ") is None      # TBK_PROF_NOJ: intervals only (the main chain without the junction branch beside it)
_cov = ctx.coverage
ctx.coverage = lambda v, **kw: _cov(v, want_junc=WJ, **kw)
c = ctx.coverage(view, out=bufs, raw=True)
ctx.set_profiling(True)
acc = {}
t0 = time.perf_counter()
for _ in range(reps):
    c = ctx.coverage(view, out=bufs, raw=True)
    for k, (ms, ln) in ctx.kernel_times().items():
        a = acc.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += ln
wall = (time.perf_counter() - t0) / reps
ctx.set_profiling(False)
t0 = time.perf_counter()
for _ in range(reps):
    c = ctx.coverage(view, out=bufs, raw=True)
wall2 = (time.perf_counter() - t0) / reps
b_cov = g["n_groups"] * 12 + 4 * view.n_cigar_ops + 16 * c["span_bases"] + 16 * c["n_intervals"]
print("records %d  span %d  intervals %d  junctions %d  bases %d  alg bytes %.3f GB" % (view.n_records, c["span_bases"], c["n_intervals"], c["n_junctions"], c["n_bases"], b_cov / 1e9))
print("coverage call: %.3f ms (profiling on: %.3f ms)" % (wall2 * 1e3, wall * 1e3))
for k, (ms, ln) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-20s %8.3f ms/step  %3d launches  %8.1f us each" % (k, ms / reps, ln // reps, 1e3 * ms / ln))
ms, ln = acc["cov_tile"]
print("cov_tile: %.1f us -> %.1f GB/s algorithmic = %.3f of 8 TB/s" % (1e3 * ms / ln, b_cov / (ms / ln * 1e-3) / 1e9, b_cov / (ms / ln * 1e-3) / 8e12))
