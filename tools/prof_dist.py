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
    def pack_partials(s, loc, ff, cap):
        torch.cuda.synchronize(); t0=time.perf_counter()
        r = ctx.pack_partials(loc, ff, cap, out=s.b.setdefault("p",{})); torch.cuda.synchronize(); print("  pack %.3f ms"%((time.perf_counter()-t0)*1e3)); return r
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

# ---- per-op timing of the torch glue -----------------------------------------------------------------------
import collections
acc = collections.defaultdict(float)
def wrap(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] += (time.perf_counter() - t0) * 1e3
        return r
    return staticmethod(w)
for name in ("to_i64", "u32_to_i64", "zeros", "full", "arange", "cummax", "cumsum", "searchsorted", "repeat", "stack", "cat", "host", "where", "bincount", "as_dtype"):
    setattr(D._TT, name, wrap(name, getattr(D._TT, name)))
import builtins
torch.cuda.synchronize(); t0 = time.perf_counter()
r = dist.run_loopback(c, [dt], [0], want_coverage=True, device_chain=True)
torch.cuda.synchronize(); print("instrumented step %.3f ms" % ((time.perf_counter() - t0) * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("   %-14s %.3f ms" % (k, v))
