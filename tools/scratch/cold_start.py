import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tiebrush_amd import api, synth_dev
t0=time.perf_counter(); d=synth_dev.make_tile_device(32, 1000000, "c2", device="cuda:0"); torch.cuda.synchronize(); print("gen", time.perf_counter()-t0)
for sz in (1<<30, 5<<30):
    t0=time.perf_counter(); x=torch.empty(sz, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize(); print("torch alloc", sz>>30, "GB", time.perf_counter()-t0); del x
torch.cuda.empty_cache()
ctx=api.Context(0)
for i in range(4):
    t0=time.perf_counter(); g=ctx.collapse(d, raw=True); dt=time.perf_counter()-t0; print("collapse call", i, "%.1f ms"%(dt*1e3), g["n_groups"])
ctx2=api.Context(0)
o=ctx2.make_opts(defer_yd=True)
for i in range(3):
    t0=time.perf_counter(); g=ctx2.collapse(d, opts=o, raw=True); t1=time.perf_counter(); ctx2.finish_yd(); t2=time.perf_counter(); print("deferred: main %.1f ms  yd wait %.1f ms"%((t1-t0)*1e3,(t2-t1)*1e3))
