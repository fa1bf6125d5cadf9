#!/usr/bin/env python3
"""bench.py — headline benchmark of the tiebrush/tiecov hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
tbk_collapse_tile (k-way merge order, grouping, YC/YX/YD) -> device chain (tbk_groups_to_cov_in) -> tbk_coverage_tile
(bedgraph intervals + junctions) of the collapsed records.

Workload (BASELINE.json `configs`):
  N = 1   configs[2] = the largest single-GPU configuration: 64 synthetic sorted BAMs x 5M 100-bp reads, --clip collapse,
          then tiecov -c -j of the result                                                       (--profile c3 defaults)
  N > 1   configs[3]'s shape: 32 files x 2M reads PER RANK (256 files over 8 GPUs), default CIGAR-only collapse; inside the
          timed step the ranks agree on bundle-aligned coordinate cuts (all-gather of sampled keys, all-reduce rounds),
          shuffle the passing records by coordinate (all-to-all over RCCL/xGMI) and each collapses + covers its own range
          over ALL files (tiebrush_amd/dist.py); the shuffle of step i + 1 overlaps the collapse / tiecov / YD of step i (two tail
          contexts, every collective issued by the main thread).  Weak scaling: per-rank input is fixed.
The tile is generated on the GPU (tiebrush_amd/synth_dev.py) before the timed region.

Prints ONE JSON line (rank 0).  value = input alignment records collapsed per second, whole job, inputs resident in HBM.
Extra objects: `roofline` (dominant kernel, HIP events on the launch stream), `roofline_coverage` (cov_tile),
`kernel_path_host_to_host` (the same step with the SoA starting in pinned host memory and every result ending there:
H2D + D2H inside the clock, SURVEY.md §8d — reported, never `value`), `cpu_baseline` (the CPU oracle, 1 thread, gcc -O2
and the reference's shipped -O0, on a bounded coordinate window of the same tile).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured stream copy)

WORKLOADS = {   # profile -> (files per GPU, reads per file, collapse options, description)
    "c2": (2, 1_000_000, {}, "default CIGAR-only"),
    "c3": (64, 5_000_000, dict(strategy="clip"), "--clip"),
    "c4": (32, 2_000_000, {}, "default CIGAR-only"),
    "c5": (128, 1_000_000, dict(strategy="exon", max_nh=5, min_qual=1), "--exon -N 5 -Q 1"),
}
ORACLE_KW = {"c2": {}, "c3": dict(strategy=2), "c4": {}, "c5": dict(strategy=3, max_nh=5, min_qual=1)}
SYNTH_PROFILE = {"c2": "c2", "c3": "c3", "c4": "c2", "c5": "c5"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--profile", default=None, choices=sorted(WORKLOADS), help="default: c3 at N=1, c4 at N>1")
    ap.add_argument("--files-per-gpu", type=int, default=None)
    ap.add_argument("--reads-per-file", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=2, help="serialised per-kernel timing steps after the timed region (>= 1)")
    ap.add_argument("--contexts", type=int, default=2, help="contexts (each with its own host thread) that take the steps in turn")
    ap.add_argument("--cpu-sample-records", type=int, default=12_000_000, help="target size of the CPU-baseline window")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank (shuffle-then-collapse) path even with one rank")
    args = ap.parse_args()
    args.prof_steps = max(1, args.prof_steps)

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
    torch.cuda.set_device(local_rank)

    profile = args.profile or ("c3" if world == 1 else "c4")
    files, reads, strat, strat_name = WORKLOADS[profile]
    files = args.files_per_gpu or files
    reads = args.reads_per_file or reads

    import threading

    from tiebrush_amd import api, synth, synth_dev
    tx = synth.make_transcriptome()
    t_gen = time.perf_counter()
    dtile = synth_dev.make_tile_device(files, reads, SYNTH_PROFILE[profile], device=dev, first_file=rank * files, tx=tx)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    n_records = dtile.n_records
    n_cig_in = int(dtile.cig.numel())
    # Two contexts take the steps in turn (software pipelining of independent tiles, as a streaming host would run them):
    # the YD list machine of step i — deferred onto its context's side stream — overlaps the tiecov chain of step i and the
    # collapse of step i + 1.  Every step's YD is complete before the timed region ends.
    NCTX = max(1, args.contexts)
    ctxs = [api.Context(local_rank) for _ in range(NCTX)]
    ctx = ctxs[0]
    opts_defer = ctx.make_opts(defer_yd=True, **strat)
    cbufs2, vbufs2 = [{} for _ in range(NCTX)], [{} for _ in range(NCTX)]
    cbufs, vbufs = cbufs2[0], vbufs2[0]
    step_no = [0]

    class TailCompute:
        """what runs behind the shuffle (collapse of the owned range, device chain, tiecov, YD) on a context of its own"""

        def __init__(self, cx):
            self.cx = cx
            self.bufs = {}

        def collapse(self, tile, **kw):
            return self.cx.collapse(tile, out=self.bufs.setdefault(("c", tile.n_files), {}), **kw)

        def groups_to_cov_in(self, fin):
            return self.cx.groups_to_cov_in(fin)

        def coverage(self, view):
            return self.cx.coverage(view, out=self.bufs.setdefault("v", {}), raw=True)

        def finish_yd(self):
            self.cx.finish_yd()

    class TailHandle:
        def __init__(self):
            self.res, self.err, self.th = None, None, None

        def wait(self):
            if self.th is not None:
                self.th.join()
                self.th = None
            if self.err is not None:
                raise self.err
            return self.res

    class StitchCompute:
        """compute object of tiebrush_amd.dist.  The shuffle (merge keys, cuts, pack, unpack) runs on `ctx` in the calling
        thread, which also issues every collective; what follows it — no collective inside — is handed over (submit_tail) to
        one of two tail contexts with a thread each, so tile i's collapse / tiecov / YD overlap tile i + 1's shuffle and
        exchange.  Receive buffers alternate with the tails."""

        def __init__(self):
            self.bufs = {}
            self.tails = None                        # two contexts, created when the first tail is handed over
            self.pending = [None, None]
            self.turn = 0

        def shard_prepare(self, tile, **kw):
            return ctx.shard_prepare(tile, out=self.bufs.setdefault("sp", {}), **kw)

        def shard_probe_max(self, *a):
            return ctx.shard_probe_max(*a)

        def shard_probe_next(self, *a):
            return ctx.shard_probe_next(*a)

        def shard_pack(self, *a):
            return ctx.shard_pack(*a, out=self.bufs.setdefault("pk", {}))

        def shard_unpack(self, rows, file_off2):
            if self.pending[self.turn] is not None:      # the tail that still reads this set of receive buffers
                self.pending[self.turn].wait()
            return ctx.shard_unpack(rows, file_off2, out=self.bufs.setdefault(("up", self.turn), {}))

        def submit_tail(self, fn):
            if self.tails is None:
                self.tails = [TailCompute(api.Context(local_rank)) for _ in range(2)]
            i = self.turn
            self.turn ^= 1
            h = TailHandle()

            def run():
                try:
                    h.res = fn(self.tails[i])
                except BaseException as e:             # surfaces at wait(): a failed step fails the bench
                    h.err = e

            h.th = threading.Thread(target=run)
            h.th.start()
            self.pending[i] = h
            return h

        def drain(self):
            for h in self.pending:
                if h is not None:
                    h.wait()

    from tiebrush_amd import dist as tdist
    stitch = StitchCompute()

    def step(tile=dtile):
        if use_dist:
            # bundle-aligned cuts (all-gather + all-reduce) -> all-to-all of the passing records over RCCL/xGMI -> one
            # collapse of the owned coordinate range over all files -> tiecov of it, everything resident in HBM
            return tdist.run_distributed(stitch, tile, rank * files, device=dev, want_coverage=True, device_chain=True, **strat)
        i = step_no[0] % NCTX
        step_no[0] += 1
        cx = ctxs[i]
        g = cx.collapse(tile, opts=opts_defer, want_coords=True, out=cbufs2[i], raw=True)   # (waits for this context's previous YD stage)
        view = cx.groups_to_cov_in(g)
        c = cx.coverage(view, out=vbufs2[i], raw=True)
        return g, c

    def drain():
        if use_dist:
            stitch.drain()                                # every tail (collapse, tiecov, YD of its tile) has finished
            return
        for cx in ctxs:
            cx.finish_yd()                                # every output of every step, YD included, is final here

    # K steps = K independent tiles.  Without collectives the two contexts are driven by two host threads (the C ABI blocks its
    # caller while a stage runs and ctypes drops the GIL meanwhile): tile i + 1's collapse runs beside tile i's tiecov chain and
    # YD stage — what a streaming host with two workers does.  Launch-bound workloads (config 2) gain most; config 3 keeps the GPU
    # busy either way.
    last = [None] * NCTX

    def run_steps(k):
        if use_dist:
            for _ in range(k):
                last[0] = step()
            return

        errs = []

        def worker(i, cnt):
            cx = ctxs[i]
            try:
                for _ in range(cnt):
                    gq = cx.collapse(dtile, opts=opts_defer, want_coords=True, out=cbufs2[i], raw=True)
                    view = cx.groups_to_cov_in(gq)
                    cq = cx.coverage(view, out=vbufs2[i], raw=True)
                    last[i] = (gq, cq)
            except BaseException as e:                    # a failed step fails the bench, never a silent short count
                errs.append(e)

        th = [threading.Thread(target=worker, args=(i, (k + NCTX - 1 - i) // NCTX)) for i in range(NCTX)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if errs:
            raise errs[0]

    run_steps(args.warmup)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_used_gb = round((total_b - free_b) / 1e9, 1)      # tile + every context's arena and outputs, after the timed region
    if use_dist:
        r = last[0].wait()
        g, c = {"n_passed": r.n_passed_local, "n_groups": r.n_groups}, r.coverage
    else:
        g, c = next(x for x in last if x is not None)
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

        for _ in range(args.prof_steps):   # the same calls as a timed step, one after the other: every kernel is measured with
            # the GPU to itself (in the timed loop the YD stage and the next tile's collapse run beside the tiecov chain)
            gg = ctx.collapse(dtile, opts=opts_defer, want_coords=True, out=cbufs, raw=True)
            take("collapse")
            ctx.finish_yd()
            take("collapse")               # the deferred YD stage belongs to tbk_collapse_tile
            view = ctx.groups_to_cov_in(gg)
            take("chain")                  # tiebrush -> tiecov device chain (representatives gathered into tiecov's input view)
            cc = ctx.coverage(view, out=vbufs, raw=True)
            take("coverage")
        ctx.set_profiling(False)
        # algorithmic bytes; the profiled steps are rank 0's local collapse + coverage
        b_collapse = gg["n_passed"] * 16 + 4 * n_cig_in
        ncig_cov = int(view.n_cigar_ops)
        b_cov = gg["n_groups"] * 12 + 4 * ncig_cov + 16 * cc["span_bases"] + 16 * cc["n_intervals"]
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_%s_%dx%d.json" % (profile, files, reads))
        if os.path.exists(tpath):          # PMC passes (FETCH_SIZE / WRITE_SIZE, corrected) of this same workload, per launch
            traffic = json.load(open(tpath)).get("bytes_per_launch", {})

        def roofline(stage, name, alg_bytes):
            ms, ln = acc[(stage, name)]
            per_launch_ms = ms / ln
            launches_per_step = ln / args.prof_steps
            achieved = (alg_bytes / launches_per_step) / (per_launch_ms * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic.get(name),
                    "avg_launch_us": round(per_launch_ms * 1e3, 2), "launches_per_step": launches_per_step,
                    "measured": "HIP events on the launch stream, profiling steps after the timed region, calls serialised (kernel alone on the GPU)",
                    "algorithmic_bytes_per_step": int(alg_bytes)}

        tot = {k: v[0] / args.prof_steps for k, v in acc.items()}
        dom = max(tot, key=tot.get)
        roof["roofline"] = roofline(dom[0], dom[1], b_collapse if dom[0] == "collapse" else b_cov)
        if ("coverage", "cov_tile") in acc:
            roof["roofline_coverage"] = roofline("coverage", "cov_tile", b_cov)
        cdom = max((k for k in tot if k[0] == "collapse"), key=tot.get)
        roof["roofline_collapse"] = roofline("collapse", cdom[1], b_collapse)
        roof["kernel_ms_per_step"] = {"%s/%s" % k: round(v, 4) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
        roof["launches_per_step"] = int(sum(v[1] for v in acc.values()) / args.prof_steps)
        roof["gpu_kernel_ms_per_step_total"] = round(sum(tot.values()), 4)

    # ---- kernel path, pinned host -> pinned host: H2D of the SoA and D2H of every result inside the clock ----
    host_path = None
    if rank == 0 and not use_dist and not args.no_host_path:
        names = ("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig")
        hin = {k: torch.empty(getattr(dtile, k).shape, dtype=getattr(dtile, k).dtype, pin_memory=True) for k in names}
        for k in names:
            hin[k].copy_(getattr(dtile, k))
        torch.cuda.synchronize()
        from dataclasses import replace
        stage = replace(dtile, **{k: torch.empty_like(getattr(dtile, k)) for k in names})
        hout = {}

        def host_step():
            for k in names:
                getattr(stage, k).copy_(hin[k], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            step_no[0] = 0
            gq, cq = step(stage)
            drain()
            ng, ni, nj = gq["n_groups"], cq["n_intervals"], cq["n_junctions"]
            outs = [(cbufs[k], ng) for k in ("rep", "yc", "yx", "yd", "g_start", "g_end")]
            outs += [(vbufs[k], ni) for k in ("iv_tid", "iv_start", "iv_end", "iv_val")]
            outs += [(vbufs[k], nj) for k in ("j_tid", "j_start", "j_end", "j_strand", "j_val")]
            nbytes = 0
            for i, (t, cnt) in enumerate(outs):
                if i not in hout or hout[i].numel() < cnt:
                    hout[i] = torch.empty(max(int(cnt * 1.1), 1), dtype=t.dtype, pin_memory=True)
                hout[i][:cnt].copy_(t[:cnt], non_blocking=True)
                nbytes += cnt * t.element_size()
            torch.cuda.synchronize()
            return gq["n_passed"], nbytes

        host_step()
        reps = 3
        t1 = time.perf_counter()
        for _ in range(reps):
            npass, out_bytes = host_step()
        hdt = (time.perf_counter() - t1) / reps
        in_bytes = sum(hin[k].numel() * hin[k].element_size() for k in names)
        host_path = {"value": round(npass / hdt, 1), "unit": "records/s", "ms_per_step": round(hdt * 1e3, 3),
                     "h2d_bytes": int(in_bytes), "d2h_bytes": int(out_bytes), "reps": reps,
                     "note": "SoA in pinned host memory -> groups, intervals and junctions in pinned host memory; PCIe inside the clock"}
        del hin, stage, hout

    # ---- CPU baseline: the oracle (literal single-threaded restatement of the reference) on a bounded coordinate window ----
    cpu = None
    if rank == 0 and not use_dist and not args.no_cpu_baseline:
        from oracle import oracle_ffi as orc
        okw = ORACLE_KW[profile]
        # window: the first w bases of chr3 in every file, w sized so that about --cpu-sample-records records fall inside
        frac = min(1.0, args.cpu_sample_records / max(n_records, 1))
        if frac >= 1.0:
            sample = synth_dev.tile_to_host(dtile)
            wdesc = "the whole tile"
        else:
            n_t2 = int((dtile.tid == 2).sum())
            w = int(synth.REF_LENS[2] * min(1.0, args.cpu_sample_records / max(n_t2, 1)))
            sample = synth_dev.tile_to_host(dtile, window=(2, 0, w))
            wdesc = "all %d files restricted to chr3:0-%d" % (files, w)

        def cpu_leg(opt, budget_s):
            reps = 0
            t1 = time.perf_counter()
            while True:
                og = orc.collapse(sample, opt=opt, **okw)
                oc = orc.coverage(synth.collapsed_to_cov_input(sample, og), opt=opt)
                reps += 1
                if time.perf_counter() - t1 > budget_s or reps >= 20:
                    break
            return og, oc, og["n_passed"] * reps / (time.perf_counter() - t1), reps

        og, oc, v2, r2 = cpu_leg("O2", 10.0)
        _, _, v0, r0 = cpu_leg("O0", 10.0)
        # the GPU path on the same sample must agree with the oracle (counts here; tests/ compare every array)
        sg = ctx.collapse(api.to_device(sample, dev), **strat)
        assert sg["n_passed"] == og["n_passed"] and sg["n_groups"] == og["n_groups"], "GPU/oracle disagree on the bench sample"
        cpu = {"value": round(v2, 1), "unit": "records/s", "cores": 1, "kind": "port",
               "sample": "%s: %d records -> %d groups, collapse+coverage on SoA, gcc -O2, %d repetition(s)" %
                         (wdesc, sample.n_records, og["n_groups"], r2),
               "value_O0": round(v0, 1), "note_O0": "same code at -O0 -g, how the reference ships (CMakeLists.txt:49), %d repetition(s)" % r0,
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
                                   % (profile, files, reads, strat_name),
                       "records_per_gpu": int(n_records), "groups_out": int(n_groups), "parallelism": "files-per-rank x%d" % world,
                       "resident": "SoA in HBM before the timed region", "generated_on_device_s": round(t_gen, 2),
                       "contexts": NCTX, "hbm_in_use_gb": hbm_used_gb},
            "bases_per_s": round(tot_bases * args.steps / dt, 1),
            "tiecov": {"bases_covered_per_step": int(n_bases), "bundle_span_bases": int(span), "intervals": int(n_iv), "junctions": int(n_j)},
        }
        line.update(roof)
        if host_path is not None:
            line["kernel_path_host_to_host"] = host_path
        e2e_path = os.path.join(ROOT, "profiles", "r2_e2e_32x1M.json")
        if os.path.exists(e2e_path):   # the command lines end to end (BAM files -> BAM file): measured by tools/e2e_bench.py, not in this run
            e = json.load(open(e2e_path))
            line["end_to_end"] = {"value": e["records_per_s_end_to_end"], "unit": "records/s", "workload": e["workload"],
                                  "wall_s": e["tiebrush_wall_s"], "host_decode_wall_s": e.get("host_decode_wall_s"),
                                  "source": "profiles/r2_e2e_32x1M.json (tools/e2e_bench.py on the same kind of box; process start, BGZF both ways, PCIe inside the clock)"}
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
