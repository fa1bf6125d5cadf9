#!/bin/bash
python - <<P
import sys,os,subprocess,tempfile,time,shutil
sys.path.insert(0,".")
import torch
from tiebrush_amd import synth, synth_dev
d=tempfile.mkdtemp(prefix="tbk_dd_",dir="/tmp")
tile=synth_dev.tile_to_host(synth_dev.make_tile_device(32,1000000,"c2",device="cuda:0"))
torch.cuda.empty_cache()
paths=synth.write_bams_fast(tile,os.path.join(d,"in"),seq=True)
del tile
for env in ({"TBK_DEVICE_DECODE":"1"},{}):
    for _ in range(2):
        t=time.time()
        r=subprocess.run(["tiebrush_amd/_build/tiebrush","-o",os.path.join(d,"o.bam")]+paths,capture_output=True,text=True,env=dict(os.environ,TBK_TIMING="1",**env))
        print(env, round(time.time()-t,3))
    print("\n".join(l for l in r.stderr.split("\n") if "ms" in l))
shutil.rmtree(d)
P
