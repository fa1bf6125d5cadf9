import os, sys, time, zlib
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np
from tiebrush_amd import api, bamio
from test_gpu_bgzf import _bgzf
raw = open(os.path.join(ROOT,"tests","golden","t12.bam"),"rb").read()
pay = bamio.bgzf_decompress(raw)
print("fixture payload", len(pay), "compressed", len(raw), "ratio %.2f" % (len(pay)/len(raw)))
reps = max(1, (1<<30)//len(pay))
comp = raw[:-28] * reps        # drop the EOF member, replicate whole members
ctx = api.Context(0)
ctx.bgzf_inflate(raw)
ctx.set_profiling(True)
t0=time.perf_counter(); out = ctx.bgzf_inflate(comp); dt=time.perf_counter()-t0
kt = ctx.kernel_times()
print("members ~%d  payload %.2f GB  call %.1f ms (H2D+D2H incl.)  kernel %s" % (len(comp)//20000, len(out)/1e9, dt*1e3, kt))
ms = kt["bgz_inflate"][0]
print("inflate kernel: %.1f ms -> %.1f GB/s of payload" % (ms, len(out)/ms/1e6))
assert out[:len(pay)] == pay and out[-len(pay):] == pay
for reps2 in (1, 16, 64):
    c2 = raw[:-28] * reps2
    ctx.bgzf_inflate(c2)
    t0=time.perf_counter(); ctx.bgzf_inflate(c2); dt=time.perf_counter()-t0
    print("reps", reps2, "kernel", ctx.kernel_times().get("bgz_inflate"), "call ms %.2f" % (dt*1e3))
