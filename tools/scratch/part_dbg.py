"""debug: owner-side PART window path vs sort path on the same partial tile"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tiebrush_amd import api, dist, synth
from tiebrush_amd.soa import SoATile
from dist_helpers import split_tile

tile = synth.make_tile(7, 20000, "c3", n_loci=800)
tiles, first = split_tile(tile, 3)
ctx = api.Context(0)
rows_all, cig_all, cnt = [], [], []
for r in range(3):
    dt = api.to_device(tiles[r], "cuda:0")
    fin = ctx.collapse(dt, strategy="clip", want_coords=True, want_effend=True)
    key, emax, bad = ctx.partial_keys(dt, fin)
    rows, cigw, tab = ctx.partial_pack(dt, fin, key, None, 1, first[r])
    th = tab.cpu().numpy()
    rows_all.append(rows.clone()); cig_all.append(cigw[:int(th[0, 2])].clone()); cnt.append(int(th[0, 1]))
rows = torch.cat(rows_all); cig = torch.cat(cig_all)
A = ctx.partial_unpack(rows)
fo2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
A = {k: v.clone() for k, v in A.items()}
t2 = SoATile(n_files=3, file_off=fo2, tbmerged=np.ones(3, np.uint8), tid=A["tid"], pos=A["pos"], flag=A["flag"], mapq=A["mapq"], strand=A["strand"],
             nh=A["nh"], cig_off=A["cig_off"], cig=cig, yc_in=A["yc_in"], yx_in=A["yx_in"], yd_in=A["yd_in"], prio_hi=A["prio_hi"], prio_lo=A["prio_lo"])
res = {}
for path in ("sort", "window"):
    os.environ["TBK_PATH"] = path
    res[path] = api.to_numpy(ctx.collapse(t2, strategy="clip", want_coords=True, want_rec_group=True, keep_supplementary=True, keep_secondary=True))
a, b = res["sort"], res["window"]
print("groups", a["n_groups"], b["n_groups"])
for k in ("yc", "yx", "yd", "g_start", "g_end", "rep", "rec_group"):
    print(k, np.array_equal(a[k], b[k]))
bad = np.nonzero(a["rep"] != b["rep"])[0]
print("bad reps", len(bad))
ph, pl = A["prio_hi"].cpu().numpy(), A["prio_lo"].cpu().numpy()
rg = a["rec_group"]
for g in bad[:8]:
    mem = np.nonzero(rg == g)[0]
    print("group", g, "members", mem.tolist(), "prio_hi", ph[mem].tolist(), "prio_lo", [(int(x) >> 32, int(x) & 0xFFFFFFFF) for x in pl[mem]],
          "sort rep", a["rep"][g], "window rep", b["rep"][g])
