import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from tiebrush_amd import api, synth_dev
dt = synth_dev.make_tile_device(64, 5_000_000, "c3", device="cuda:0")
ctx = api.Context(0)
g = ctx.collapse(dt, strategy="clip")
rep = g["rep"].to(torch.int64) & 0xFFFFFFFF
st = g["g_start"].to(torch.int64); en = g["g_end"].to(torch.int64); tid = dt.tid[rep].to(torch.int64)
key = (tid << 32) | st
ekey = (tid << 32) | en
rm = torch.cummax(ekey, 0)[0]
head = torch.ones_like(st, dtype=torch.bool); head[1:] = key[1:] > rm[:-1]
bid = torch.cumsum(head.to(torch.int64), 0) - 1
nb = int(bid[-1]) + 1
b_start = st[head]
b_end = torch.zeros(nb, dtype=torch.int64, device=st.device).scatter_reduce_(0, bid, en, "amax", include_self=False)
span = b_end - b_start + 1
b_off = torch.cumsum(span, 0) - span
cs = b_off[bid] + (st - b_start[bid])
t = cs // 8192
cnt = torch.bincount(t)
c = cnt.cpu().numpy()
print("tiles", len(c), "records", len(st), "mean", c.mean(), "p50", np.median(c), "p99", np.percentile(c, 99), "p99.9", np.percentile(c, 99.9), "max", c.max())
srt = np.sort(c)[::-1]
print("top 10 tiles:", srt[:10])
print("records in tiles > 8192:", c[c > 8192].sum(), "tiles:", (c > 8192).sum(), " >32768:", (c > 32768).sum())
