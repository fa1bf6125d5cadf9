#!/bin/bash
mkdir -p gpurun_out/devp
timeout -k 10 600 python - > gpurun_out/devp/out.txt 2>&1 <<P
import sys, os, time, tempfile, shutil, subprocess
sys.path.insert(0, ".")
import torch
from tiebrush_amd import synth, synth_dev
d = tempfile.mkdtemp(prefix="tbk_dp_", dir="/tmp")
tile = synth_dev.tile_to_host(synth_dev.make_tile_device(32, 1000000, "c2", device="cuda:0"))
torch.cuda.empty_cache()
paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=True)
del tile
exe = os.path.join("tiebrush_amd", "_build", "tiebrush")
for tag, env in (("device", dict(TBK_DEVICE_DECODE="1", TBK_DEV_TWICE="1")), ("device", dict(TBK_DEVICE_DECODE="1")), ("hybrid", dict(TBK_HYBRID="1")), ("host", dict(TBK_HYBRID="0"))):
    out = os.path.join(d, "out.bam")
    t = time.time()
    r = subprocess.run([exe, "-o", out] + paths, capture_output=True, text=True, env=dict(os.environ, TBK_TIMING="1", **env))
    print("=====", tag, "wall %.3f" % (time.time() - t))
    print('\n'.join(l for l in r.stderr.split('\n') if l.startswith('device path') or l.startswith('hybrid path') or l.startswith('host path')))
shutil.rmtree(d)
P
grep -v amdgpu.ids gpurun_out/devp/out.txt | tail -n 60
