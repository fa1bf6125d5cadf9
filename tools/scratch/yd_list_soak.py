"""One-off soak of the window path's round-3 forms against the oracle: YD items placed by list (<= 64 files) and by the radix split
(> 64), file masks, the verification list, the tiecov view built from g_key — on synthetic tiles of random shape with unstranded
reads mixed in.  usage: yd_list_soak.py [tiles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_fuzz import _cmp
from oracle import oracle_ffi as orc
from tiebrush_amd import api, synth
ctx = api.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(424242)
t0 = time.time()
KW = {"c2": (dict(), dict()), "c3": (dict(strategy="clip"), dict(strategy=2)), "c5": (dict(strategy="exon", max_nh=5, min_qual=1), dict(strategy=3, max_nh=5, min_qual=1))}
for it in range(N):
    files = int(rng.choice([1, 2, 3, 5, 17, 32, 33, 63, 64, 65, 100]))
    reads = int(rng.integers(200, 4000))
    prof = str(rng.choice(["c2", "c3", "c5"]))
    tile = synth.make_tile(files, reads, prof, n_loci=int(rng.integers(5, 400)))
    tile.strand = tile.strand.copy()
    m = rng.random(len(tile.strand)) < rng.choice([0.0, 0.1, 0.5])
    tile.strand[m] = ord(".")
    kw, okw = KW[prof]
    if rng.random() < 0.3:
        kw, okw = (dict(), dict())          # default strategy on any profile
    os.environ["TBK_DEBUG"] = "path=window,raw=" + ("1" if rng.random() < 0.8 else "0")
    want = _cmp(ctx, tile, **kw)
    # device chain with the view from the keys
    dt = api.to_device(tile, "cuda:0")
    res = ctx.collapse(dt, **kw)
    cov = api.to_numpy(ctx.coverage(ctx.groups_to_cov_in(res)))
    cw = orc.coverage(synth.collapsed_to_cov_input(tile, want))
    for k in ("iv_tid", "iv_start", "iv_end", "iv_val", "j_tid", "j_start", "j_end", "j_strand", "j_val"):
        assert np.array_equal(cov[k], cw[k]), (it, k, files, reads, prof)
    if it % 20 == 19:
        print("tile", it + 1, "%.0f s" % (time.time() - t0), flush=True)
print("soak ok:", N, "tiles")
