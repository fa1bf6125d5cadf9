#!/bin/bash
python - <<P
import sys,os,time,tempfile,shutil, ctypes as C
sys.path.insert(0,".")
import torch, numpy as np
from tiebrush_amd import synth, synth_dev, api, _lib
d=tempfile.mkdtemp(prefix="tbk_dd_",dir="/tmp")
tile=synth_dev.tile_to_host(synth_dev.make_tile_device(32,1000000,"c2",device="cuda:0"))
torch.cuda.empty_cache()
paths=synth.write_bams_fast(tile,os.path.join(d,"in"),seq=True)
del tile
raw=[open(p,"rb").read() for p in paths]
ctx=api.Context(0)
o=ctx.make_opts()
for it in range(3):
    t=time.perf_counter()
    s,fo=ctx.bam_decode(raw)
    t1=time.perf_counter()
    n=int(s.n_records)
    rep=np.empty(n,np.uint32); yc=np.empty(n,np.float64); yx=np.empty(n,np.int64); yd=np.empty(n,np.int32)
    g=_lib.GroupsOut(_lib.TBK_MEM_HOST, n, rep.ctypes.data, yc.ctypes.data, yx.ctypes.data, yd.ctypes.data, None, None, None, None, None, 0, 0)
    t2=time.perf_counter()
    rc=ctx.L.tbk_collapse_tile(ctx.h, C.byref(o), C.byref(s), C.byref(g))
    t3=time.perf_counter()
    print("decode ms %.1f | out arrays %.1f | collapse (host outputs) ms %.1f rc %d groups %d" % ((t1-t)*1e3,(t2-t1)*1e3,(t3-t2)*1e3, rc, g.n_groups), flush=True)
    ctx.bam_release()
shutil.rmtree(d)
P
