#!/usr/bin/env python3
"""How much of a bench step is Python: cProfile over 200 steps of the bench loop (collapse -> chain -> coverage -> finish_yd)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiebrush_amd import api, synth
tile = synth.make_tile(2, 1_000_000, "c2")
ctx = api.Context(0)
dt = api.to_device(tile, "cuda:0")
opts = ctx.make_opts(defer_yd=True)
cb, vb = {}, {}
def step():
    g = ctx.collapse(dt, opts=opts, want_coords=True, out=cb, raw=True)
    v = ctx.groups_to_cov_in(g)
    c = ctx.coverage(v, out=vb, raw=True)
    ctx.finish_yd()
for _ in range(10): step()
t0 = time.perf_counter()
for _ in range(200): step()
print("plain loop: %.3f ms/step" % ((time.perf_counter() - t0) * 1e3 / 200))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
