"""Diagnosis: HIP's own log (AMD_LOG_LEVEL=4) of the command line around the copy that brings the representatives back."""
import atexit, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiebrush_amd import synth, synth_dev
files, reads = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp(prefix="tbk_log_", dir="/tmp")
atexit.register(shutil.rmtree, d, True)
tile = synth_dev.tile_to_host(synth_dev.make_tile_device(files, reads, "c2", device="cuda:0"))
paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=False)
del tile
log = os.path.join(ROOT, "gpurun_out", "cli_amdlog.txt")
with open(log, "w") as f:
    subprocess.run([os.path.join(ROOT, "tiebrush_amd", "_build", "tiebrush"), "-o", os.path.join(d, "out.bam")] + paths, stdout=f, stderr=f,
                   env=dict(os.environ, TBK_TIMING="1", AMD_LOG_LEVEL="4"))
lines = open(log, errors="replace").read().split("\n")
import re
t0 = None
for i, l in enumerate(lines):
    m = re.search(r": (\d+) us:", l)
    if m and t0 is None:
        t0 = int(m.group(1))
    if "HSA Copy copy_engine" in l or "Query copy engine" in l or ("hipMemcpyAsync: Returned" in l and "duration" in l and int(re.search(r"duration: (\d+)", l).group(1)) > 1000):
        ts = (int(m.group(1)) - t0) / 1000.0 if m else -1
        tid = re.search(r"tid: (0x[0-9a-f]+)", l)
        msg = l.split("] ", 1)[-1]
        msg = re.sub(r"dst=0x[0-9a-f]+, src=0x[0-9a-f]+, ", "", msg)
        print("%9.1f ms %s %s" % (ts, tid.group(1)[-5:] if tid else "", msg[:170]))
os.remove(log)
