import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tiebrush_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] == "old":
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "_old", "libtbk_r2a.so")
from test_gpu_fuzz import STRATS, SNUM, _rand_tile
from tiebrush_amd import api
from oracle import oracle_ffi as orc
ctx = api.Context(0)
rng = np.random.default_rng(31000 + 32)
for it in range(6):
    tile = _rand_tile(rng, with_tb=False)
    if it < 5: rng.choice(STRATS); rng.choice([1, 5, 2**31 - 1]); rng.choice([-1, 1, 31])
os.environ["TBK_PATH"] = "sort"
want = orc.collapse(tile, want_rec_group=True, strategy=0)
got = api.to_numpy(ctx.collapse(tile, want_rec_group=True, strategy="cigar"))
i = np.nonzero(np.asarray(got["yd"]) != np.asarray(want["yd"]))[0]
print(sys.argv[1:], "yd diffs at", i, np.asarray(got["yd"])[i], np.asarray(want["yd"])[i])
if len(i):
    g = int(i[0]); rg = np.asarray(want["rec_group"])
    mem = np.nonzero(rg == g)[0]
    fo = tile.file_of()
    co = tile.cig_off.astype(np.int64)
    print("group", g, "start/end", want["g_start"][g], want["g_end"][g], "rep", want["rep"][g], "yx", want["yx"][g])
    for m in mem:
        print("  member rec", m, "file", fo[m], "tid", tile.tid[m], "pos", tile.pos[m], "flag", tile.flag[m], "strand", chr(tile.strand[m]), "cig", ["%d%s" % (w >> 4, "MIDNSHP=XB"[w & 15]) for w in tile.cig[co[m]:co[m+1]]])
    # the groups of the same tid near it, with their samples
    tid = tile.tid[mem[0]]
    for gg in range(max(0, g - 12), min(want["n_groups"], g + 3)):
        mm = np.nonzero(rg == gg)[0]
        r = int(want["rep"][gg])
        print(" g", gg, "tid", tile.tid[r], "start", want["g_start"][gg], "end", want["g_end"][gg], "strand", chr(tile.strand[r]), "files", sorted(set(int(fo[x]) for x in mm)), "yd gpu/oracle", got["yd"][gg], want["yd"][gg], "cig", ["%d%s" % (w >> 4, "MIDNSHP=XB"[w & 15]) for w in tile.cig[co[r]:co[r+1]]])
