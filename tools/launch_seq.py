#!/usr/bin/env python3
"""Launch sequence of one bench step.  Two modes:
  launch_seq.py run [profile files reads]   (under rocprofv3 --kernel-trace): 3 serialised steps
  launch_seq.py show <kernel_trace.csv>      prints the last step's launches in start order: offset, duration, queue, name"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch
    from tiebrush_amd import api, synth_dev
    prof = sys.argv[2] if len(sys.argv) > 2 else "c2"
    nf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    nr = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    kw = {"c2": {}, "c3": dict(strategy="clip")}[prof]
    tile = synth_dev.make_tile_device(nf, nr, prof, "cuda:0")
    ctx = api.Context(0)
    for _ in range(3):
        g = ctx.collapse(tile, defer_yd=True, want_coords=True, **kw)
        c = ctx.coverage(ctx.groups_to_cov_in(g))
        ctx.finish_yd()
        torch.cuda.synchronize()
        torch.zeros(1, device="cuda:0").add_(1.0)          # marker kernel between steps
        torch.cuda.synchronize()
else:
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "CUDAFunctorOnSelf_add" in r[3] or "AUnaryFunctor<float" in r[3]]
    lo, hi = (marks[-2] + 1, marks[-1]) if len(marks) >= 2 else (0, len(rows))
    t0 = rows[lo][0]
    for b, e, q, k in rows[lo:hi]:
        print("%9.1f us +%7.1f q%-3s %s" % ((b - t0) / 1e3, (e - b) / 1e3, q, k[:110]))
    print("launches:", hi - lo, "span %.1f us" % ((rows[hi - 1][1] - t0) / 1e3))
