#!/bin/bash
# the decode paths of the tiebrush command line on 32 files x 1 M reads with SEQ / QUAL: the CLI parity test, then wall time and
# phase times (TBK_TIMING) of host-only, hybrid (the GPU and the cores a share of the files each) at several shares, device-only
mkdir -p gpurun_out/hyb
timeout -k 10 500 python -m pytest tests/test_gpu_cli.py -x -q -k decode_paths > gpurun_out/hyb/pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/hyb/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 700 python - > gpurun_out/hyb/e2e.txt 2>&1 <<P
import sys, os, time, tempfile, shutil, subprocess
sys.path.insert(0, ".")
import torch
from tiebrush_amd import synth, synth_dev
d = tempfile.mkdtemp(prefix="tbk_hyb_", dir="/tmp")
tile = synth_dev.tile_to_host(synth_dev.make_tile_device(32, 1000000, "c2", device="cuda:0"))
torch.cuda.empty_cache()
paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=True)
del tile
print("input bytes", sum(os.path.getsize(p) for p in paths))
exe = os.path.join("tiebrush_amd", "_build", "tiebrush")
for tag, env in (("host", dict(TBK_HYBRID="0")), ("hybrid 40", dict(TBK_HYBRID="1")), ("hybrid 55", dict(TBK_HYBRID="1", TBK_HYBRID_SHARE="55")),
                 ("hybrid 70", dict(TBK_HYBRID="1", TBK_HYBRID_SHARE="70")), ("device", dict(TBK_DEVICE_DECODE="1")),
                 ("host", dict(TBK_HYBRID="0")), ("hybrid 40", dict(TBK_HYBRID="1")), ("device", dict(TBK_DEVICE_DECODE="1"))):
    out = os.path.join(d, "out.bam")
    t = time.time()
    r = subprocess.run([exe, "-o", out] + paths, capture_output=True, text=True, env=dict(os.environ, TBK_TIMING="1", **env))
    dt = time.time() - t
    ph = [l for l in r.stderr.split("\n") if l.startswith("host path") or l.startswith("hybrid path") or l.startswith("device path ms")]
    print("%-10s wall %.3f s rc %d | %s" % (tag, dt, r.returncode, ph[-1] if ph else r.stderr[-300:]), flush=True)
shutil.rmtree(d)
P
grep -v amdgpu.ids gpurun_out/hyb/e2e.txt
