"""One-off soak: synthetic tiles of random shape (files, reads, loci, profile) through the forced window path (raw) and the default
path against the oracle, every output array."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_fuzz import _cmp
from tiebrush_amd import api, synth
ctx = api.Context(0)
rng = np.random.default_rng(4242)
n = 0
for it in range(int(sys.argv[1])):
    prof = str(rng.choice(["c2", "c3", "c5"]))
    files = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65, 100, 300]))
    reads = int(rng.integers(200, 200000 // files + 300))
    loci = int(rng.choice([3, 20, 200, 2000]))
    tile = synth.make_tile(files, reads, prof, n_loci=loci)
    kw = {"c2": {}, "c3": dict(strategy="clip"), "c5": dict(strategy="exon", max_nh=5, min_qual=1)}[prof]
    for path in ("window", None):
        if path: os.environ["TBK_DEBUG"] = "path=" + path
        else: os.environ.pop("TBK_DEBUG", None)
        try:
            _cmp(ctx, tile, **kw); n += 1
        except AssertionError as e:
            print("FAIL", it, prof, files, reads, loci, path, str(e)[:200], flush=True)
print("synth soak ok:", n)
