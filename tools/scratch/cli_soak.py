"""One-off soak of the command line: random small inputs as BAM files through `tiebrush` (host decode with tiny tiles, host decode
with one tile, device decode) against the oracle."""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_fuzz import _rand_tile
from oracle import oracle_ffi as orc
from tiebrush_amd import bamio, synth
BIN = os.path.join(ROOT, "tiebrush_amd", "_build", "tiebrush")
d = tempfile.mkdtemp(prefix="tbk_soak_")
nfail = n = 0
for seed in range(int(sys.argv[1])):
    rng = np.random.default_rng(77000 + seed)
    tile = _rand_tile(rng, with_tb=False)
    tile.tid = np.where(tile.tid > 2, 2, tile.tid)
    if tile.n_records == 0: continue
    paths = synth.write_bams(tile, os.path.join(d, "s%d_" % seed))
    fo = tile.file_of()
    for strat, flag, okw in (("cigar", [], {}), ("clip", ["-P"], dict(strategy=2)), ("exon", ["-E"], dict(strategy=3))):
        want = orc.collapse(tile, **okw)
        for mode, env in (("tiny tiles", dict(TBK_TILE_RECORDS="3", TBK_DEVICE_DECODE="0")), ("one tile", dict(TBK_DEVICE_DECODE="0")), ("device decode", {})):
            out = os.path.join(d, "o.bam")
            r = subprocess.run([BIN, "-o", out] + flag + paths, capture_output=True, text=True, env=dict(os.environ, **env))
            n += 1
            if r.returncode != 0:
                print("FAIL rc", seed, strat, mode, r.stderr[-200:], flush=True); nfail += 1; continue
            o = bamio.read_bam(out)
            ok = o.n == want["n_groups"] and [int(x) for x in o.yd] == [int(x) for x in want["yd"]] and [int(x) for x in o.yx] == [int(x) for x in want["yx"]] \
                and [float(x) for x in o.yc] == [float(np.float32(x)) for x in want["yc"]]
            if ok:   # the representative: same file-local record (QNAME r<file>_<index>)
                names = [b"r%d_%d" % (int(fo[g]), int(g) - int(tile.file_off[int(fo[g])])) for g in want["rep"]]
                ok = [bytes(x) for x in o.qname] == names if hasattr(o, "qname") else True
            if not ok:
                print("FAIL", seed, strat, mode, o.n, want["n_groups"], flush=True); nfail += 1
print("cli soak:", n, "runs,", nfail, "failures")
