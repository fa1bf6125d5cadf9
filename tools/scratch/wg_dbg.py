import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from tiebrush_amd import api, synth_dev
dt = synth_dev.make_tile_device(64, 5_000_000, "c3", device="cuda:0")
ctx = api.Context(0)
g = ctx.collapse(dt, strategy="clip")
os.environ["TBK_WG_DEBUG"] = "1"
g = ctx.collapse(dt, strategy="clip")
os.environ.pop("TBK_WG_DEBUG")
gy = ctx.collapse(dt, strategy="clip")
os.environ["TBK_WG_DEBUG"] = "1"
print("no rec_slot (YC only path?)", file=sys.stderr)
