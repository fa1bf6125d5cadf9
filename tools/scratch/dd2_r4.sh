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
print("compressed GB", sum(len(r) for r in raw)/1e9)
ctx=api.Context(0)
for it in range(3):
    ctx.set_profiling(it==2)
    t=time.perf_counter()
    s,fo=ctx.bam_decode(raw)
    dt=time.perf_counter()-t
    print("bam_decode wall ms", round(dt*1e3,1), "records", s.n_records)
    if it==2:
        for k,(ms,ln) in sorted(ctx.kernel_times().items(), key=lambda kv:-kv[1][0]): print("   %-24s %8.2f ms %d launches"%(k,ms,ln))
    ctx.bam_release()
shutil.rmtree(d)
P
