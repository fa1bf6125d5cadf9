#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary when the caller hands over HOST buffers (TBK_MEM_HOST): the same config-2 tile
as bench.py, numpy arrays in pageable memory in, numpy arrays out.  Never what bench.py reports as `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiebrush_amd import api, synth

tile = synth.make_tile(2, 1_000_000, "c2")
ctx = api.Context(0)
for _ in range(3):
    r = ctx.collapse(tile)
t0 = time.perf_counter()
K = 10
for _ in range(K):
    r = ctx.collapse(tile)
dt = (time.perf_counter() - t0) / K
inb = sum(a.nbytes for a in (tile.tid, tile.pos, tile.flag, tile.mapq, tile.strand, tile.nh, tile.cig_off, tile.cig))
outb = r["n_groups"] * (4 + 8 + 8 + 4 + 4 + 4)
print("host-pointer collapse: %.2f ms per tile, %.1f M records/s, %.1f MB in + %.1f MB out over PCIe" %
      (dt * 1e3, tile.n_records / dt / 1e6, inb / 1e6, outb / 1e6))
