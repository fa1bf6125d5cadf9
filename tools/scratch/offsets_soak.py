"""One-off soak of the window path's partition (wg_offsets_stream_k: blocks of eight chunks, carried bounds) against the oracle: synthetic tiles of
random shape — runs shorter and longer than a chunk, a block, many runs per chunk — raw and compacted form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_fuzz import _cmp
from tiebrush_amd import api, synth
n = 0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 515)
for raw in ("1", "0"):
    os.environ["TBK_DEBUG"] = "path=window,raw=" + raw
    ctx = api.Context(0)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        prof = ["c2", "c3", "c5"][it % 3]
        files = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65, 130, 300]))
        reads = int(rng.integers(50, max(60, 400000 // files)))
        tile = synth.make_tile(files, reads, prof, n_loci=int(rng.integers(3, 600)))
        kw = {"c2": {}, "c3": dict(strategy="clip"), "c5": dict(strategy="exon", max_nh=5, min_qual=1)}[prof]
        _cmp(ctx, tile, **kw); n += 1
        if it % 5 == 0:
            print(raw, it, files, reads, "ok"); sys.stdout.flush()
    ctx.close()
print("soak ok:", n, "comparisons")
