"""Where the time of one multi-rank step goes at world = 1 (loopback, one context, every phase alone on the GPU): host wall per
phase (synchronised at every mark) and HIP-event kernel times of the owner's stages.
usage: python tools/prof_dist.py [files reads profile]      (default: config 4's per-rank shape, 32 x 2M)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiebrush_amd import api, synth, synth_dev, dist

files = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
prof = sys.argv[3] if len(sys.argv) > 3 else "c2"
kw = {"c2": {}, "c3": dict(strategy="clip"), "c5": dict(strategy="exon", max_nh=5, min_qual=1)}[prof]
dt = synth_dev.make_tile_device(files, reads, prof, device="cuda:0")
ctx = api.Context(0)


class M:
    def __init__(s):
        s.b, s.t, s.acc, s.kt = {}, None, {}, {}
    def __getattr__(s, name):
        return getattr(ctx, name)
    def _k(s, stage):
        for k, (ms, ln) in ctx.kernel_times().items():
            a = s.kt.setdefault((stage, k), [0.0, 0]); a[0] += ms; a[1] += ln
    def collapse(s, t, **kw):
        r = ctx.collapse(t, out=s.b.setdefault(("c", t.n_files), {}), **kw); s._k("collapse k=%d" % t.n_files); return r
    def coverage(s, v):
        r = ctx.coverage(v, out=s.b.setdefault("v", {}), raw=True); s._k("coverage"); return r
    def groups_to_cov_in(s, f):
        r = ctx.groups_to_cov_in(f); s._k("chain"); return r
    def partial_keys(s, t, f):
        r = ctx.partial_keys(t, f, out=s.b.setdefault("pk", {})); s._k("keys"); return r
    def partial_pack(s, *a, **kw):
        r = ctx.partial_pack(*a, out=s.b.setdefault("pp", {}), **kw); s._k("pack"); return r
    def partial_reduce(s, *a, **kw):
        r = ctx.partial_reduce(*a, out=s.b.setdefault("pr", {}), **kw); s._k("reduce"); return r
    def partial_unpack(s, rows):
        r = ctx.partial_unpack(rows, out=s.b.setdefault("pu", {})); s._k("unpack"); return r
    def mark(s, name):
        torch.cuda.synchronize(); now = time.perf_counter()
        if s.t is not None: s.acc[name] = s.acc.get(name, 0.0) + (now - s.t) * 1e3
        s.t = now


m = M()
R = 6
for it in range(R + 2):
    if it == 2:
        m.acc, m.kt = {}, {}
        ctx.set_profiling(True)
    torch.cuda.synchronize(); m.t = time.perf_counter()
    res = dist.run_loopback(m, [dt], [0], want_coverage=True, device_chain=True, **kw)
print("records %d -> local groups / partials %d" % (dt.n_records, res[0].n_partials_received))
print("phases (ms, host wall, synchronised, profiling on):", {k: round(v / R, 3) for k, v in m.acc.items()}, "sum %.3f" % (sum(m.acc.values()) / R))
st = {}
for (stage, k), (ms, ln) in m.kt.items():
    st.setdefault(stage, []).append((ms / R, ln / R, k))
for stage, rows in st.items():
    rows.sort(reverse=True)
    print("%-16s kernels %.3f ms, %d launches: " % (stage, sum(r[0] for r in rows), sum(r[1] for r in rows)) + ", ".join("%s %.3f" % (r[2], r[0]) for r in rows[:14]))
ctx.set_profiling(False)
for it in range(R + 2):
    if it == 2:
        m.acc = {}
    torch.cuda.synchronize(); m.t = time.perf_counter()
    dist.run_loopback(m, [dt], [0], want_coverage=True, device_chain=True, **kw)
print("phases (ms, host wall, synchronised, profiling off):", {k: round(v / R, 3) for k, v in m.acc.items()}, "sum %.3f" % (sum(m.acc.values()) / R))
