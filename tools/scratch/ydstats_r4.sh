#!/bin/bash
# one collapse of config 3's tile with the instrumented yd_wave_k (gpurun_exp/yds/libtbk.so: -DYD_STATS), categories printed to stderr
cp gpurun_exp/yds/libtbk.so tiebrush_amd/_build/libtbk.so
mkdir -p gpurun_out/yds
timeout -k 10 400 python - > gpurun_out/yds/out.txt 2>&1 <<P
import sys, ctypes
sys.path.insert(0, ".")
import torch
from tiebrush_amd import synth_dev, api, _lib
dt = synth_dev.make_tile_device(64, 5000000, "c3", device="cuda:0")
ctx = api.Context(0)
opts = ctx.make_opts(strategy="clip")
fin = ctx.collapse(dt, opts=opts, want_coords=True)
torch.cuda.synchronize()
L = _lib.load()
f = ctypes.CDLL(_lib.LIB_PATH).tbk_yd_stats_dump
f()
print("groups", fin.n_groups if hasattr(fin, "n_groups") else fin)
P
cat gpurun_out/yds/out.txt | tail -n 30
