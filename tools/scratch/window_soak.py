"""One-off soak of the window path (raw and compacted) against the oracle on many random small tiles and a few synthetic ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_fuzz import STRATS, _cmp, _rand_tile
from tiebrush_amd import api, synth
ctx = api.Context(0)
n = 0
for raw in ("1", "0"):
    os.environ["TBK_DEBUG"] = "path=window,raw=" + raw   # (the binding forwards a changed TBK_DEBUG to the live context)
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        rng = np.random.default_rng(31000 + seed)
        for _ in range(6):
            tile = _rand_tile(rng, with_tb=False)
            for strat in STRATS:
                _cmp(ctx, tile, strategy=strat); n += 1
            _cmp(ctx, tile, strategy=str(rng.choice(STRATS)), keep_secondary=True, keep_supplementary=True,
                 max_nh=int(rng.choice([1, 5, 2**31 - 1])), min_qual=int(rng.choice([-1, 1, 31]))); n += 1
    for seed in range(6):
        prof = ["c2", "c3", "c5"][seed % 3]
        files = [3, 17, 65, 130][seed % 4]
        tile = synth.make_tile(files, 4000 + 1000 * seed, prof, n_loci=50 + 40 * seed, seed=77 + seed) if "seed" in synth.make_tile.__code__.co_varnames else synth.make_tile(files, 4000 + 1000 * seed, prof, n_loci=50 + 40 * seed)
        kw = {"c2": {}, "c3": dict(strategy="clip"), "c5": dict(strategy="exon", max_nh=5, min_qual=1)}[prof]
        _cmp(ctx, tile, **kw); n += 1
print("soak ok:", n, "comparisons")
