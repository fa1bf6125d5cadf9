#!/usr/bin/env python3
"""bench.py — headline benchmark of the tiebrush/tiecov hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: tbk_collapse_tile (k-way merge order, grouping, YC/YX/YD) -> device chain
(tbk_groups_to_cov_in) -> tbk_coverage_tile (bedgraph intervals + junctions) of the collapsed
records.  Workload at N=1 = BASELINE.json configs[1]: 2 synthetic sorted BAMs x 1M 100-bp reads,
default CIGAR-only collapse.  With N>1 every rank owns its own 2 input files (weak scaling: the
N input streams shard per rank); inside the timed step the ranks agree on
bundle-aligned coordinate cuts (all-gather of sampled keys, all-reduce rounds), shuffle the passing
records by coordinate (all-to-all over RCCL) and each collapses + covers its own range over ALL files
(tiebrush_amd/dist.py).

Prints ONE JSON line (rank 0).  value = input alignment records collapsed per second, whole job.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--files-per-gpu", type=int, default=2)
    ap.add_argument("--reads-per-file", type=int, default=1_000_000)
    ap.add_argument("--profile", default="c2", choices=["c2", "c3", "c5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=5)
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank (shuffle-then-collapse) path even with one rank")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # TBK_BENCH_BACKEND=gloo is a test hook: several ranks may then share one GPU (collectives staged through the
        # host), so this script's multi-rank path can be exercised on a 1-GPU box; measured runs use RCCL, one GPU per rank
        backend = os.environ.get("TBK_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = "cuda:%d" % local_rank

    from tiebrush_amd import api, synth
    strat = {"c2": {}, "c3": dict(strategy="clip"), "c5": dict(strategy="exon", max_nh=5, min_qual=1)}[args.profile]
    tx = synth.make_transcriptome()
    tile = synth.make_tile(args.files_per_gpu, args.reads_per_file, args.profile, first_file=rank * args.files_per_gpu, tx=tx)
    ctx = api.Context(local_rank)
    dtile = api.to_device(tile, dev)
    opts = ctx.make_opts(**strat)
    opts_defer = ctx.make_opts(defer_yd=True, **strat)   # YD list machine overlaps the (YD-independent) tiecov chain
    cbufs, vbufs = {}, {}

    class StitchCompute:
        """compute object of tiebrush_amd.dist: the same context, output buffers reused per call site"""

        def __init__(self):
            self.bufs = {}

        def collapse(self, tile, **kw):
            return ctx.collapse(tile, out=self.bufs.setdefault(("c", tile.n_files, "prio" if tile.prio_hi is not None else ""), {}), **kw)

        def groups_to_cov_in(self, fin):
            return ctx.groups_to_cov_in(fin)

        def shard_prepare(self, tile, **kw):
            return ctx.shard_prepare(tile, out=self.bufs.setdefault("sp", {}), **kw)

        def shard_probe_max(self, *a):
            return ctx.shard_probe_max(*a)

        def shard_probe_next(self, *a):
            return ctx.shard_probe_next(*a)

        def shard_pack(self, *a):
            return ctx.shard_pack(*a, out=self.bufs.setdefault("pk", {}))

        def shard_unpack(self, rows, file_off2):
            return ctx.shard_unpack(rows, file_off2, out=self.bufs.setdefault("up", {}))

        def finish_yd(self):
            ctx.finish_yd()

        def coverage(self, view):
            return ctx.coverage(view, out=self.bufs.setdefault("v", {}), raw=True)

    from tiebrush_amd import dist as tdist
    stitch = StitchCompute()

    def step():
        if use_dist:
            # bundle-aligned cuts (all-gather + all-reduce) -> all-to-all of the passing records over RCCL/xGMI -> one
            # collapse of the owned coordinate range over all files -> tiecov of it, everything resident in HBM
            r = tdist.run_distributed(stitch, dtile, rank * args.files_per_gpu, device=dev, want_coverage=True,
                                      device_chain=True, **strat)
            return ({"n_passed": r.n_passed_local, "n_groups": r.n_groups}, r.coverage)
        g = ctx.collapse(dtile, opts=opts_defer, want_coords=True, out=cbufs, raw=True)
        view = ctx.groups_to_cov_in(g)
        c = ctx.coverage(view, out=vbufs, raw=True)
        ctx.finish_yd()                                   # every output of the step, YD included, is final here
        return g, c

    for _ in range(args.warmup):
        g, c = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g, c = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_passed, n_groups = g["n_passed"], g["n_groups"]
    n_bases, span, n_iv, n_j = c["n_bases"], c["span_bases"], c["n_intervals"], c["n_junctions"]
    stats = torch.tensor([dt, float(n_passed), float(n_bases)], dtype=torch.float64,
                         device=dev if os.environ.get("TBK_BENCH_BACKEND", "nccl") == "nccl" else "cpu")
    if use_dist:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        tot_records, tot_bases = float(sm[1]), float(sm[2])
    else:
        tot_records, tot_bases = float(n_passed), float(n_bases)

    # ---- per-kernel durations (HIP events on the launch stream) -> roofline of the dominant kernel ----
    roof = {}
    if rank == 0:
        ctx.set_profiling(True)
        acc = {}

        def take(stage):
            for k, (ms, ln) in ctx.kernel_times().items():
                a = acc.setdefault((stage, k), [0.0, 0])
                a[0] += ms
                a[1] += ln

        for _ in range(args.prof_steps):   # the same step as the timed loop (YD stage overlapping the tiecov chain)
            gg = ctx.collapse(dtile, opts=opts_defer, want_coords=True, out=cbufs, raw=True)
            take("collapse")
            view = ctx.groups_to_cov_in(gg)
            cc = ctx.coverage(view, out=vbufs, raw=True)
            take("coverage")
            ctx.finish_yd()
            take("collapse")               # the deferred YD stage belongs to tbk_collapse_tile
        ctx.set_profiling(False)
        ncig_in = int(tile.cig.shape[0])
        # algorithmic bytes
        # (the profiled steps are rank 0's local collapse + coverage; their own counts price the bytes)
        b_collapse = gg["n_passed"] * 16 + 4 * ncig_in
        ncig_cov = int(view.n_cigar_ops)
        b_cov = gg["n_groups"] * 12 + 4 * ncig_cov + 16 * cc["span_bases"] + 16 * cc["n_intervals"]

        def roofline(stage, name, alg_bytes):
            ms, ln = acc[(stage, name)]
            per_launch_ms = ms / ln
            launches_per_step = ln / args.prof_steps
            achieved = (alg_bytes / launches_per_step) / (per_launch_ms * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_us": round(per_launch_ms * 1e3, 2), "launches_per_step": launches_per_step,
                    "algorithmic_bytes_per_step": int(alg_bytes)}

        tot = {k: v[0] / args.prof_steps for k, v in acc.items()}
        dom = max(tot, key=tot.get)
        roof["roofline"] = roofline(dom[0], dom[1], b_collapse if dom[0] == "collapse" else b_cov)
        roof["roofline_coverage"] = roofline("coverage", "cov_tile", b_cov)
        cdom = max((k for k in tot if k[0] == "collapse"), key=tot.get)
        roof["roofline_collapse"] = roofline("collapse", cdom[1], b_collapse)
        roof["kernel_ms_per_step"] = {"%s/%s" % k: round(v, 4) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
        roof["gpu_kernel_ms_per_step_total"] = round(sum(tot.values()), 4)

    # ---- CPU baseline: the oracle (literal single-threaded port of the reference) on rank 0's shard ----
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle_ffi as orc
        okw = {"c2": {}, "c3": dict(strategy=2), "c5": dict(strategy=3, max_nh=5, min_qual=1)}[args.profile]
        reps = 0
        t1 = time.perf_counter()
        while True:
            og = orc.collapse(tile, **okw)
            oc = orc.coverage(synth.collapsed_to_cov_input(tile, og))
            reps += 1
            if time.perf_counter() - t1 > 10.0 or reps >= 20:
                break
        cdt = time.perf_counter() - t1
        assert og["n_passed"] == n_passed and (use_dist or og["n_groups"] == n_groups), "GPU/oracle disagree on the bench workload"
        assert use_dist or (oc["n_intervals"] == n_iv and oc["n_junctions"] == n_j)
        cpu = {"value": round(og["n_passed"] * reps / cdt, 1), "unit": "records/s", "cores": 1, "kind": "port",
               "sample": "rank-0 shard (%d files x %d reads) collapse+coverage on SoA, %d repetitions, gcc -O2" %
                         (args.files_per_gpu, args.reads_per_file, reps),
               "host_cores_available": os.cpu_count()}

    if rank == 0:
        line = {
            "metric": "input alignment records/sec collapsed (tiebrush) + bases/sec covered (tiecov)",
            "value": round(tot_records * args.steps / dt, 1),
            "unit": "records/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": "%s: %d synthetic sorted BAMs x %d 100bp reads per GPU, %s collapse + tiecov -c -j of the result"
                                   % (args.profile, args.files_per_gpu, args.reads_per_file,
                                      {"c2": "default CIGAR-only", "c3": "--clip", "c5": "--exon -N 5 -Q 1"}[args.profile]),
                       "records_per_gpu": int(tile.n_records), "groups_out": int(n_groups), "parallelism": "files-per-rank x%d" % world,
                       "resident": "SoA in HBM before the timed region"},
            "bases_per_s": round(tot_bases * args.steps / dt, 1),
            "tiecov": {"bases_covered_per_step": int(n_bases), "bundle_span_bases": int(span), "intervals": int(n_iv), "junctions": int(n_j)},
        }
        line.update(roof)
        if cpu is not None:
            line["cpu_baseline"] = cpu
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio: flush that first so the JSON line is the LAST line on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
