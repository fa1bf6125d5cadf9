#!/bin/bash
python - <<P
import sys,os,time,tempfile,shutil
sys.path.insert(0,".")
import torch, numpy as np
from tiebrush_amd import synth, synth_dev, api
d=tempfile.mkdtemp(prefix="tbk_dd_",dir="/tmp")
tile=synth_dev.tile_to_host(synth_dev.make_tile_device(32,1000000,"c2",device="cuda:0"))
torch.cuda.empty_cache()
paths=synth.write_bams_fast(tile,os.path.join(d,"in"),seq=True)
del tile
raw=[open(p,"rb").read() for p in paths]
ctx=api.Context(0)
for it in range(3):
    t=time.perf_counter()
    s,fo=ctx.bam_decode(raw)
    t1=time.perf_counter()
    ctx.set_profiling(it==2)
    g=ctx.collapse_struct(s, 32)
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print("decode ms %.1f collapse ms %.1f groups %s" % ((t1-t)*1e3,(t2-t1)*1e3, g["n_groups"] if g else None), flush=True)
    if it==2:
        kt=ctx.kernel_times()
        print("kernel sum ms %.1f launches %d" % (sum(v[0] for v in kt.values()), sum(v[1] for v in kt.values())))
        for k,(ms,ln) in sorted(kt.items(), key=lambda kv:-kv[1][0])[:8]: print("   %-24s %8.2f ms %d launches"%(k,ms,ln))
    ctx.set_profiling(False)
    ctx.bam_release()
shutil.rmtree(d)
P
