import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tiebrush_amd import api, synth, dist
import tiebrush_amd.dist as D
tile = synth.make_tile(2, 1000000, "c2")
ctx = api.Context(0)
dt = api.to_device(tile, "cuda:0")
class C:
    def __init__(s): s.b={}
    def collapse(s, t, **kw):
        torch.cuda.synchronize(); t0=time.perf_counter()
        r = ctx.collapse(t, out=s.b.setdefault(("c",t.n_files,t.prio_hi is not None),{}), **kw)
        torch.cuda.synchronize(); print("  collapse n=%d %.3f ms"%(t.n_records if not hasattr(t.tid,'numel') else t.tid.numel(), (time.perf_counter()-t0)*1e3)); return r
    def groups_to_cov_in(s, f):
        torch.cuda.synchronize(); t0=time.perf_counter()
        r= ctx.groups_to_cov_in(f); torch.cuda.synchronize(); print("  g2c %.3f ms"%((time.perf_counter()-t0)*1e3)); return r
    def _timed(s, name, fn, *a, **k):
        torch.cuda.synchronize(); t0=time.perf_counter()
        r = fn(*a, **k); torch.cuda.synchronize(); print("  %s %.3f ms"%(name, (time.perf_counter()-t0)*1e3)); return r
    def shard_prepare(s, tile, **kw): return s._timed("prepare", ctx.shard_prepare, tile, out=s.b.setdefault("sp",{}), **kw)
    def shard_probe_max(s, *a): return s._timed("probe_max", ctx.shard_probe_max, *a)
    def shard_probe_next(s, *a): return s._timed("probe_next", ctx.shard_probe_next, *a)
    def shard_pack(s, *a): return s._timed("pack", ctx.shard_pack, *a, out=s.b.setdefault("pk",{}))
    def shard_unpack(s, rows, fo): return s._timed("unpack", ctx.shard_unpack, rows, fo, out=s.b.setdefault("up",{}))
    def finish_yd(s):
        ctx.finish_yd()
    def coverage(s, v):
        torch.cuda.synchronize(); t0=time.perf_counter()
        r= ctx.coverage(v, out=s.b.setdefault("v",{}), raw=True); torch.cuda.synchronize(); print("  cov %.3f ms"%((time.perf_counter()-t0)*1e3)); return r
c=C()
for it in range(4):
    torch.cuda.synchronize(); t0=time.perf_counter()
    r = dist.run_loopback(c, [dt], [0], want_coverage=True, device_chain=True)
    torch.cuda.synchronize(); print("step %.3f ms"%((time.perf_counter()-t0)*1e3))

# ---- phase timing (host clock, synchronised at every mark) ------------------------------------------------------
class M:
    def __init__(s): s.b = {}; s.t = None; s.acc = {}
    def __getattr__(s, name):
        return getattr(ctx, name)
    def collapse(s, t, **kw): return ctx.collapse(t, out=s.b.setdefault(("c", t.n_files), {}), **kw)
    def coverage(s, v): return ctx.coverage(v, out=s.b.setdefault("v", {}), raw=True)
    def shard_prepare(s, tile, **kw): return ctx.shard_prepare(tile, out=s.b.setdefault("sp", {}), **kw)
    def shard_pack(s, *a): return ctx.shard_pack(*a, out=s.b.setdefault("pk", {}))
    def shard_unpack(s, rows, fo): return ctx.shard_unpack(rows, fo, out=s.b.setdefault("up", {}))
    def mark(s, name):
        torch.cuda.synchronize(); now = time.perf_counter()
        if s.t is not None: s.acc[name] = s.acc.get(name, 0.0) + (now - s.t) * 1e3
        s.t = now
m = M()
R = 10
for it in range(R + 2):
    if it == 2: m.acc = {}
    torch.cuda.synchronize(); m.t = time.perf_counter()
    dist.run_loopback(m, [dt], [0], want_coverage=True, device_chain=True)
print("phases (ms, synchronised):", {k: round(v / R, 3) for k, v in m.acc.items()}, "sum %.3f" % (sum(m.acc.values()) / R))
