#!/usr/bin/env python3
"""Soak of the YD stage's forms against the oracle (GPU box): random synthetic tiles of random shape — 1-64 files (items placed by list)
and 65-100 files (the radix split), shallow and deep loci, the three read models, every strategy — through the forced window path and the
default path, with the chains split as in production, all through yd_wave_k, all through yd_lane_k and all through yd_run_k.  Every
output array bit for bit.  usage: soak_yd.py [tiles [first seed]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KEYS = ("rep", "yc", "yx", "yd", "g_start", "g_end")
STRAT = {"cigar": 0, "clip": 2, "exon": 3}


def main():
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad = 0
    forms = [dict(), dict(yd_wave_min="1"), dict(yd_wave_min=str(1 << 30)), dict(yd_literal="1")]
    for it in range(n_tiles):
        rng = np.random.default_rng(seed0 + it)
        k = int(rng.integers(1, 65)) if rng.random() < 0.8 else int(rng.integers(65, 101))
        reads = int(rng.integers(50, 3000))
        profile = ["c2", "c3", "c5"][int(rng.integers(0, 3))]
        loci = int(rng.choice([3, 10, 40, 200]))
        strat = ["cigar", "clip", "exon"][int(rng.integers(0, 3))]
        kw = dict(max_nh=5, min_qual=1) if (profile == "c5" and rng.random() < 0.5) else {}
        tile = synth.make_tile(k, reads, profile, n_loci=loci, seed_base=int(rng.integers(0, 1 << 30)))
        want = orc.collapse(tile, strategy=STRAT[strat], **kw)
        form = forms[it % len(forms)]
        for path in ("window", None):
            dbg = dict(form)
            if path:
                dbg["path"] = path
            os.environ["TBK_DEBUG"] = ",".join("%s=%s" % kv for kv in dbg.items())
            ctx = api.Context(0)
            try:
                got = api.to_numpy(ctx.collapse(api.to_device(tile, "cuda:0"), strategy=strat, want_coords=True, **kw))
            finally:
                ctx.close()
            ok = got["n_groups"] == want["n_groups"] and all(np.array_equal(np.asarray(got[q]), np.asarray(want[q])) for q in KEYS)
            if not ok:
                bad += 1
                print("MISMATCH seed %d k %d reads %d %s loci %d %s %s form %s path %s" % (seed0 + it, k, reads, profile, loci, strat, kw, form, path), flush=True)
        if it % 50 == 49:
            print("%d tiles, %d mismatches, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
    print("soak_yd: %d tiles x 2 paths, %d mismatches, %.0f s" % (n_tiles, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
