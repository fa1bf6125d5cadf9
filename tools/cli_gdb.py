"""Diagnosis: the tiebrush command line on a synthetic whole-input run under rocgdb (a backtrace when it dies)."""
import atexit, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiebrush_amd import synth, synth_dev
files, reads = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp(prefix="tbk_gdb_", dir="/tmp")
atexit.register(shutil.rmtree, d, True)
tile = synth_dev.tile_to_host(synth_dev.make_tile_device(files, reads, "c2", device="cuda:0"))
paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=False)
del tile
cmd = [os.path.join(ROOT, "tiebrush_amd", "_build", "tiebrush"), "-o", os.path.join(d, "out.bam")] + paths
r = subprocess.run(["/opt/rocm/bin/rocgdb", "-batch", "-ex", "run", "-ex", "bt", "-ex", "info threads", "--args"] + cmd, capture_output=True, text=True,
                   env=dict(os.environ, TBK_TIMING="1"))
print(r.stdout[-6000:])
print(r.stderr[-3000:])
