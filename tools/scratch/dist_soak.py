"""One-off soak of the multi-rank path (loopback driver, device-resident tiles, tbk_shard_* kernels) on small adversarial tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_fuzz import STRATS, _rand_tile
from test_gpu_dist import DeviceCompute
from dist_helpers import split_tile, check_against_flat, STRAT
from oracle import oracle_ffi as orc
from tiebrush_amd import api, dist, synth
comp = DeviceCompute()
n = 0
for seed in range(int(sys.argv[1])):
    rng = np.random.default_rng(52000 + seed)
    for _ in range(4):
        tile = _rand_tile(rng, tiecov_safe=True, with_tb=False)
        if tile.n_records == 0: continue
        for strat in ("cigar", "clip", "exon"):
            world = int(rng.integers(1, tile.n_files + 1))
            flat = orc.collapse(tile, strategy=STRAT[strat])
            flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
            tiles, first = split_tile(tile, world)
            dtiles = [api.to_device(t, "cuda:0") for t in tiles]
            try:
                res = dist.run_loopback(comp, dtiles, first, strategy=strat, want_coverage=True, device_chain=True)
                for r in res:
                    for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
                        v = getattr(r, f)
                        setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
                check_against_flat(res, tile, flat, flat_cov)
            except AssertionError as e:
                print("FAIL seed", seed, strat, "world", world, "files", tile.n_files, "n", tile.n_records, str(e)[:200], flush=True)
            n += 1
print("dist soak done:", n)
