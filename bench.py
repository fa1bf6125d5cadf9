#!/usr/bin/env python3
"""bench.py — headline benchmark of the tiebrush/tiecov hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
tbk_collapse_tile (k-way merge order, grouping, YC/YX/YD) -> device chain (tbk_groups_to_cov_in) -> tbk_coverage_tile
(bedgraph intervals + junctions) of the collapsed records.

Workload (BASELINE.json `configs`):
  N = 1   configs[2] = the largest single-GPU configuration: 64 synthetic sorted BAMs x 5M 100-bp reads, --clip collapse,
          then tiecov -c -j of the result                                                       (--profile c3 defaults)
  N > 1   configs[3], BASELINE's scaling workload: ONE fixed job of 256 synthetic sorted BAMs x 2M reads, default collapse, its files
          split over the ranks (256 / N each; N = 8: 32 per GPU as configs[3] says) — `"scaling": "strong"`.  `--scaling weak` gives
          every rank the N = 1 workload instead (c3 per GPU; or `--profile c4 --scaling weak`: 32 files x 2M per rank), and
          `--gpus 1 --profile c4 --scaling strong` runs the whole 256-file job on one GPU (the strong-scaling base line).  Inside
          the timed step every
          rank collapses its own files (the plain single-GPU path), the ranks agree on bundle-aligned coordinate cuts
          (all-gather of sampled group keys, all-reduce rounds), exchange one 48-byte row per LOCAL GROUP plus its CIGAR
          (all-to-all over RCCL/xGMI), and each reduces the partials of its range by key and covers it (tiebrush_amd/dist.py,
          partials_collapse).  Tile i + 1's local collapse (a worker thread, two contexts in turn) overlaps tile i's
          exchange / reduce / tiecov (main thread, which issues every collective).  After the timed region every rank also
          times the plain single-GPU step on its own tile (`plain_ms_per_step`): the same workload without the exchange.
The tile is generated on the GPU (tiebrush_amd/synth_dev.py) before the timed region.

Prints ONE JSON line (rank 0).  value = input alignment records collapsed per second, whole job, inputs resident in HBM.
Extra objects, all measured in this run unless they say otherwise: `roofline` (dominant kernel, HIP events on the launch
stream), `roofline_coverage` (cov_tile: median / min / max over >= 10 serialised launches), `kernel_path_host_to_host` (the same
step with the SoA starting in pinned host memory and every result ending there: H2D + D2H inside the clock, SURVEY.md §8d —
reported, never `value`), `end_to_end` (the `tiebrush` command line on 32 x 1M-read BAM files written to /tmp, run as a child
process before this process touches the GPU: process start, BGZF both ways, tagging and PCIe inside the clock), `cpu_baseline`
(the CPU oracle: 1 thread at gcc -O2 and at the reference's shipped -O0, and a tiewrap-style multi-process line).
"""
import argparse
import contextlib
import json
import os
import subprocess
import sys
import time

# Every context drives three HIP streams (main, YD, junctions) and the bench runs two or three contexts: with the runtime's default of four
# hardware queues, streams that should overlap share a queue and run one behind the other.  Eight queues: 18.5 -> 18.3 ms per step on
# config 3, on every run (round 4's measurements, DESIGN_HISTORY.md §6).  Read by the HIP runtime when it starts, so it is set before anything loads it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
STRONG_JOB_FILES = {"c2": 2, "c3": 64, "c4": 256, "c5": 1024}   # the whole job of --scaling strong (BASELINE.json configs)
ORACLE_KW = {"c2": {}, "c3": dict(strategy=2), "c4": {}, "c5": dict(strategy=3, max_nh=5, min_qual=1)}
SYNTH_PROFILE = {"c2": "c2", "c3": "c3", "c4": "c2", "c5": "c5"}


def host_memory_budget():
    """bytes this process may use: min(cgroup limit, MemAvailable)"""
    lim = None
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(p).read().strip()
            if v.isdigit():
                lim = int(v)
                break
        except OSError:
            pass
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    c = [x for x in (lim, avail) if x]
    return min(c) if c else 16 << 30


def cpu_budget():
    """worker processes worth starting: the CPU affinity / cgroup quota, not the machine's core count"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_batch_worker(args):
    """tiewrap-style batch (tiewrap.py:96-126): collapse the files of one batch with the oracle, return the collapsed records as a
    TieBrush-merged SoA (spawned process: numpy + the oracle only, never the GPU)"""
    shm_dir, b, okw = args
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import oracle_ffi as orc
    from tiebrush_amd.soa import SoATile
    ld = lambda name: np.load(os.path.join(shm_dir, "b%d_%s.npy" % (b, name)), mmap_mode="r")
    fo = np.asarray(ld("file_off"))
    t = SoATile(n_files=len(fo) - 1, file_off=fo, tbmerged=np.zeros(len(fo) - 1, np.uint8), tid=ld("tid"), pos=ld("pos"), flag=ld("flag"),
                mapq=ld("mapq"), strand=ld("strand"), nh=ld("nh"), cig_off=ld("cig_off"), cig=ld("cig"))
    g = orc.collapse(t, **okw)
    rep = g["rep"].astype(np.int64)
    co = np.asarray(t.cig_off).astype(np.int64)
    nc = co[rep + 1] - co[rep]
    off = np.concatenate([[0], np.cumsum(nc)])
    idx = np.repeat(co[rep] - off[:-1], nc) + np.arange(int(off[-1]))
    out = dict(tid=np.asarray(t.tid)[rep], pos=np.asarray(t.pos)[rep], flag=np.asarray(t.flag)[rep], mapq=np.asarray(t.mapq)[rep],
               strand=np.asarray(t.strand)[rep], nh=np.asarray(t.nh)[rep], cig_off=off.astype(np.uint32), cig=np.asarray(t.cig)[idx],
               yc=g["yc"].astype(np.float32).astype(np.float64), yx=g["yx"], yd=g["yd"].astype(np.int64))
    for k, v in out.items():
        np.save(os.path.join(shm_dir, "o%d_%s.npy" % (b, k)), v)
    return b, int(g["n_passed"]), int(g["n_groups"])


def _cpu_warm(_):
    import numpy  # noqa: F401
    sys.path.insert(0, ROOT)
    from oracle import oracle_ffi as orc
    orc.lib("O2")
    return os.getpid()


def run_e2e_leg(args):
    """the `tiebrush` command line end to end, as a child process tree that ends before this process touches the GPU"""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_leg.py"), "--files", str(args.e2e_files), "--reads", str(args.e2e_reads),
           "--runs", str(args.e2e_runs)] + ([] if args.no_cpu_baseline else ["--cpu-baseline"])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
        line = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "tools/e2e_leg.py failed (%d): %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
        return json.loads(line[-1])
    except Exception as e:       # the leg is a report, never the headline: a failure is stated in the line, not hidden
        return {"error": repr(e)}


def launch_ranks(n):
    """Start one child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them) and wait.
    The parent initialises no GPU and imports no torch.  A rank that fails ends the others (their exact process groups, never a
    pattern) and the parent returns non-zero; rank 0 prints the one JSON line on the inherited stdout."""
    import signal
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   TBK_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        stop_all()
        raise SystemExit(128 + signum)
    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                sys.stderr.write("bench.py: rank %d ended with status %d: stopping the other ranks\n" % (r, c))
                stop_all()
                t_end = time.time() + 10
                while time.time() < t_end and any(p.poll() is None for p in procs):
                    time.sleep(0.1)
                stop_all(signal.SIGKILL)
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; without WORLD_SIZE in the environment N > 1 starts the N rank processes itself (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--profile", default=None, choices=sorted(WORKLOADS), help="default: c3 at N = 1, c4 (BASELINE's scaling workload) at N > 1")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong: one fixed job (c4: 256 files x 2M reads) split over the ranks; weak: the profile's per-GPU shape on every rank.  "
                         "Default: strong with c4, weak otherwise")
    ap.add_argument("--files-per-gpu", type=int, default=None)
    ap.add_argument("--reads-per-file", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--host-subtiles", type=int, default=12, help="sub-tiles the host -> host leg cuts the tile into (bundle boundaries)")
    ap.add_argument("--host-contexts", type=int, default=3, help="contexts (a host thread each) of the host -> host leg")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-files", type=int, default=32)
    ap.add_argument("--e2e-reads", type=int, default=1_000_000)
    ap.add_argument("--e2e-runs", type=int, default=3)
    ap.add_argument("--prof-steps", type=int, default=2, help="serialised per-kernel timing steps after the timed region (>= 1)")
    ap.add_argument("--cov-prof-reps", type=int, default=10, help="serialised coverage calls behind roofline_coverage's median")
    ap.add_argument("--contexts", type=int, default=0, help="contexts (each with its own host thread) that take the steps in turn; 0: by the tile's size")
    ap.add_argument("--cpu-sample-records", type=int, default=0, help="size of the CPU-baseline sample (0: the whole tile when host memory allows)")
    ap.add_argument("--cpu-procs", type=int, default=0, help="worker processes of the tiewrap-style CPU line (0: the CPU quota, at most 16)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank (group-partials) path even with one rank")
    ap.add_argument("--dist-mode", default="partials", choices=["partials", "shuffle"])
    ap.add_argument("--cut-search", default="lists", choices=["lists", "rounds"],
                    help="partials protocol: cuts chosen on the device from gathered bundle lists (one read-back per step), or walked in all-reduce rounds")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="with --force-dist on one GPU: R virtual ranks in this process (loopback collectives), every one with the per-rank workload: "
                         "what the R-way cut search, pack and reduce do, without R GPUs; reported in `emulated_world`, not in `value`")
    args = ap.parse_args()
    args.prof_steps = max(1, args.prof_steps)

    # `python bench.py --gpus N` with no WORLD_SIZE in the environment starts its N ranks itself (the reference's counterpart is
    # tiewrap.py:96-126, which starts its own workers): this process never touches a GPU, it only starts and reaps the children.
    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = world
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: one rank per GPU, the two must agree" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist

    # ---- end to end through the command line: a child process, finished before this process initialises the GPU ----
    e2e = None
    if rank == 0 and not use_dist and not args.no_e2e:
        e2e = run_e2e_leg(args)

    import numpy as np
    import torch
    import torch.distributed as dist

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

    # N = 1: config 3 (the largest single-GPU configuration).  N > 1: config 4, the fixed 256 x 2M job split over the ranks.
    profile = args.profile or ("c3" if world == 1 else "c4")
    scaling = args.scaling or ("strong" if profile == "c4" and world > 1 else "weak")
    files, reads, strat, strat_name = WORKLOADS[profile]
    job_files = None
    if scaling == "strong":
        job_files = args.files_per_gpu * world if args.files_per_gpu else STRONG_JOB_FILES[profile]
        if job_files % world:
            raise SystemExit("--scaling strong: %d files do not split evenly over %d ranks" % (job_files, world))
        files = job_files // world
    else:
        files = args.files_per_gpu or files
    reads = args.reads_per_file or reads

    import queue
    import threading

    from tiebrush_amd import api, synth, synth_dev
    tx = synth.make_transcriptome()
    t_gen = time.perf_counter()
    dtile = synth_dev.make_tile_device(files, reads, SYNTH_PROFILE[profile], device=dev, first_file=rank * files, tx=tx)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    torch.cuda.empty_cache()                          # (the generator's temporaries go back to the device: only the tile stays)
    n_records = dtile.n_records
    n_cig_in = int(dtile.cig.numel())
    # Contexts take the steps in turn (software pipelining of independent tiles, as a streaming host would run them): the YD list
    # machine of step i — deferred onto its context's side stream — overlaps the tiecov chain of step i and the collapse of step
    # i + 1.  Every step's YD is complete before the timed region ends.
    # (0: three contexts — 17.4 ms per step on config 3 against 17.7 with two and 140 against 97 GB of HBM — while a context's arena stays
    # below a fifth of the device: two for the 512 M-record tile of config 4 as one job)
    NCTX = args.contexts if args.contexts > 0 else (3 if n_records <= 400_000_000 and not use_dist else 2)   # (a multi-rank run has its own three)
    ctxs = [api.Context(local_rank) for _ in range(NCTX)]
    ctx = ctxs[0]
    opts_defer = ctx.make_opts(defer_yd=True, **strat)
    cbufs2, vbufs2 = [{} for _ in range(NCTX)], [{} for _ in range(NCTX)]
    cbufs, vbufs = cbufs2[0], vbufs2[0]

    # TBK_BENCH_GATE=1: one collapse at a time.  Rounds 2 and 3 needed it: left alone, two contexts fell into step — both in their
    # window kernels, then both in YD, the GPU a quarter of the time with nothing but a few long YD chains on it (a rocprofv3 timeline
    # of round 3).  With round 4's kernels (the window kernel a fifth shorter and bound by its vector instructions, the
    # YD chains by their scalar ones) the contexts do better on their own: 18.3 ms per step with the gate, 17.7 without, 17.4 with
    # three contexts (round 4, DESIGN_HISTORY.md §6) — so the gate is off unless asked for.
    gate = threading.Lock() if os.environ.get("TBK_BENCH_GATE", "0") != "0" else contextlib.nullcontext()

    # group arrays: a quarter of the records (a call that needs more reports it, TBK_E2BIG, and is repeated with the need: the
    # warm-up settles the size); one group per record, the capacity that can never overflow, was 15 GB per context on config 3
    cap_groups = max(1 << 20, n_records // 4)

    def plain_step(cx, tile, cb, vb):
        cx.finish_yd()                                # (this context's previous YD stage: waited for outside the gate)
        with gate:
            g = cx.collapse(tile, opts=opts_defer, want_coords=True, out=cb, raw=True, cap_groups=cap_groups)
        view = cx.groups_to_cov_in(g)
        c = cx.coverage(view, out=vb, raw=True)
        return g, c

    last = [None] * NCTX

    def run_plain(k, tile=dtile):
        """K independent tiles, contexts driven by one host thread each (the C ABI blocks its caller while a stage runs and ctypes
        drops the GIL meanwhile): tile i + 1's collapse runs beside tile i's tiecov chain and YD stage"""
        errs = []

        def worker(i, cnt):
            torch.cuda.set_device(local_rank)             # (the current device is per thread)
            try:
                for _ in range(cnt):
                    last[i] = plain_step(ctxs[i], tile, cbufs2[i], vbufs2[i])
            except BaseException as e:                    # a failed step fails the bench, never a silent short count
                errs.append(e)

        th = [threading.Thread(target=worker, args=(i, (k + NCTX - 1 - i) // NCTX)) for i in range(NCTX)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if errs:
            raise errs[0]
        for cx in ctxs:
            cx.finish_yd()                                # every output of every step, YD included, is final here

    # ---- multi-rank step: local collapse (worker thread) -> partials exchange, reduce, tiecov (main thread) ----------------
    from tiebrush_amd import dist as tdist
    phase_ms = {}
    wire = {"wire_bytes": 0, "wire_bytes_off_rank": 0, "wire_rows": 0, "steps": 0}

    class OwnerCompute:
        """compute object of tiebrush_amd.dist for one tile: the cut search and the pack read the local groups on the context
        that produced them (`cl`, idle while the worker uses the other one); everything behind the exchange runs on `co`"""

        def __init__(self, cl, co, bufs, done_with_local):
            self.cl, self.co, self.bufs, self.done = cl, co, bufs, done_with_local
            self.t = time.perf_counter()

        def mark(self, name):
            now = time.perf_counter()
            phase_ms[name] = phase_ms.get(name, 0.0) + (now - self.t) * 1e3
            self.t = now

        def partial_keys(self, tile, fin):
            return self.cl.partial_keys(tile, fin, out=self.bufs.setdefault("pk", {}))

        def shard_probe_max(self, *a):
            return self.cl.shard_probe_max(*a)

        def shard_probe_next(self, *a):
            return self.cl.shard_probe_next(*a)

        def partial_pack(self, *a, **kw):
            return self.cl.partial_pack(*a, out=self.bufs.setdefault("pp", {}), **kw)

        # the sender's side that only queues work (tbk_partial_stage_*): no read-back before the gathered exchange table
        def partial_stage_keys(self, tile, fin, first_fidx, carry=0):
            return self.cl.partial_stage_keys(tile, fin, first_fidx, carry, out=self.bufs.setdefault("pk", {}))

        def partial_stage_cands(self, *a):
            return self.cl.partial_stage_cands(*a)

        def partial_stage_pack(self, *a, **kw):
            return self.cl.partial_stage_pack(*a, out=self.bufs.setdefault("pp", {}), **kw)

        def partial_reduce(self, rows, run_off, cig, **kw):
            self.done()                                   # the rows have left the local buffers: the worker may reuse the slot
            return self.co.partial_reduce(rows, run_off, cig, out=self.bufs.setdefault("pr", {}), **kw)

        def partial_unpack(self, rows):
            self.done()                                   # the rows have left the local buffers: the worker may reuse the slot
            return self.co.partial_unpack(rows, out=self.bufs.setdefault("pu", {}))

        # record-shuffle fallback (carried fractional YC): same contexts
        def shard_prepare(self, tile, **kw):
            return self.cl.shard_prepare(tile, out=self.bufs.setdefault("sp", {}), **kw)

        def shard_pack(self, *a):
            return self.cl.shard_pack(*a, out=self.bufs.setdefault("spk", {}))

        def shard_unpack(self, rows, fo2):
            self.done()
            return self.co.shard_unpack(rows, fo2, out=self.bufs.setdefault("sup", {}))

        def collapse(self, tile, **kw):
            return self.co.collapse(tile, out=self.bufs.setdefault(("c", tile.n_files), {}), **kw)

        def groups_to_cov_in(self, fin):
            return self.co.groups_to_cov_in(fin)

        def coverage(self, view):
            return self.co.coverage(view, out=self.bufs.setdefault("v", {}), raw=True)

    dctx = {}

    def run_dist(k):
        if not dctx:
            dctx["local"] = [api.Context(local_rank) for _ in range(2)]
            dctx["owner"] = api.Context(local_rank)
            dctx["owner"].use_torch_stream()      # the owner's kernels are ordered with the collectives by the stream, not by host waits
            dctx["prev_nj"] = 0
            dctx["lbufs"] = [{}, {}]
            dctx["obufs"] = {}
            dctx["opts"] = dctx["local"][0].make_opts(defer_yd=True, **strat)
        cl, co = dctx["local"], dctx["owner"]
        ready = queue.Queue()
        free = [threading.Semaphore(1), threading.Semaphore(1)]

        def worker():
            torch.cuda.set_device(local_rank)
            try:
                fins = [None, None]
                for i in range(k):
                    s = i & 1
                    free[s].acquire()                     # tile i - 2's rows have left this slot's buffers
                    fins[s] = cl[s].collapse(dtile, opts=dctx["opts"], want_coords=True, want_effend=True, out=dctx["lbufs"][s], cap_groups=cap_groups)
                    if i > 0:
                        cl[s ^ 1].finish_yd()             # tile i - 1's YD ran beside tile i's window kernels
                        ready.put((i - 1, fins[s ^ 1]))
                cl[(k - 1) & 1].finish_yd()
                ready.put((k - 1, fins[(k - 1) & 1]))
            except BaseException as e:
                ready.put((-1, e))

        th = threading.Thread(target=worker)
        th.start()
        res = None
        try:
            for i in range(k):
                j, fin = ready.get()
                if j < 0:
                    raise fin
                assert j == i
                s = i & 1
                released = [False]

                def done(s=s, released=released):
                    if not released[0]:
                        released[0] = True
                        free[s].release()

                comp = OwnerCompute(cl[s], co, dctx["obufs"], done)
                st = {}
                kw = dict(strat)
                # (the junction counts of step i travel with step i + 1's first gather — the `carry` word —: the numbering offsets of a
                # step's junctions are known one step later, and one gather behind the last step settles the last)
                extra = dict(local=fin, stats=st, cut_search=args.cut_search, carry=dctx["prev_nj"], junction_gather=False) if args.dist_mode == "partials" else {}
                res = tdist.run_distributed(comp, dtile, rank * files, device=dev, want_coverage=True, device_chain=True, mode=args.dist_mode, **extra, **kw)
                done()
                if res.coverage is not None:
                    dctx["prev_nj"] = int(res.coverage["n_junctions"])
                for kk in ("wire_bytes", "wire_bytes_off_rank", "wire_rows", "collectives", "host_syncs"):
                    wire[kk] = wire.get(kk, 0) + st.get(kk, 0)
                wire["cut_rounds_max"] = max(wire.get("cut_rounds_max", 0), st.get("cut_rounds", 0))
                wire["steps"] += 1
            if args.dist_mode == "partials" and res is not None and res.coverage is not None and dist.is_initialized():
                njt = torch.tensor([int(res.coverage["n_junctions"])], dtype=torch.int64, device=dev if os.environ.get("TBK_BENCH_BACKEND", "nccl") == "nccl" else "cpu")
                alln = [torch.zeros_like(njt) for _ in range(world)]
                dist.all_gather(alln, njt)                 # the last step's junction counts: its numbering offsets
                res.junction_offset = int(sum(int(x) for x in alln[:rank]))
        finally:
            th.join()
        return res

    def run_steps(k):
        if use_dist:
            return run_dist(k)
        run_plain(k)
        return None

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    phase_ms.clear()
    for kk in wire:
        wire[kk] = 0
    t0 = time.perf_counter()
    dres = run_steps(args.steps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_used_gb = round((total_b - free_b) / 1e9, 1)      # tile + every context's arena and outputs, after the timed region
    if use_dist:
        g, c = {"n_passed": dres.n_passed_local, "n_groups": dres.n_groups}, dres.coverage
    else:
        g, c = next(x for x in last if x is not None)
    n_passed, n_groups = g["n_passed"], g["n_groups"]
    n_bases, span, n_iv, n_j = c["n_bases"], c["span_bases"], c["n_intervals"], c["n_junctions"]
    stats = torch.tensor([dt, float(n_passed), float(n_bases)], dtype=torch.float64,
                         device=dev if os.environ.get("TBK_BENCH_BACKEND", "nccl") == "nccl" else "cpu")
    ranks_seen = 1
    if use_dist:
        ones = torch.ones(1, dtype=torch.float64, device=stats.device)   # every rank adds one: what the collective really spanned
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones[0])))
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        tot_records, tot_bases = float(sm[1]), float(sm[2])
    else:
        tot_records, tot_bases = float(n_passed), float(n_bases)

    # ---- multi-rank: the same per-rank workload through the plain single-GPU path, after the timed region (every rank runs it,
    # nothing is exchanged): what the exchange costs, on the same tile, in the same run ----
    dist_extra = {}
    if use_dist:
        ksteps = max(2, min(args.steps, 10))
        run_plain(min(2, ksteps))
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        run_plain(ksteps)
        torch.cuda.synchronize()
        plain_dt = (time.perf_counter() - t1) / ksteps
        dist.barrier()
        pg, pc = next(x for x in last if x is not None)
        nst = max(wire["steps"], 1)
        ph = {k_: round(v / nst, 3) for k_, v in phase_ms.items()}
        dist_extra = {
            "plain_ms_per_step": round(plain_dt * 1e3, 4),
            "plain_records_per_s_per_gpu": round(pg["n_passed"] / plain_dt, 1),
            "shuffle_ms_per_step": round(sum(ph.get(k_, 0.0) for k_ in ("cuts", "pack", "exchange", "unpack", "prepare")), 3),
            "wire_bytes_per_step": int(wire["wire_bytes"] / nst),
            "wire_bytes_off_rank_per_step": int(wire["wire_bytes_off_rank"] / nst),
            "partials_per_step": int(wire["wire_rows"] / nst),
            "cut_rounds_max": int(wire.get("cut_rounds_max", 0)),
            "cut_search": args.cut_search,
            "collectives_per_step": round(wire.get("collectives", 0) / nst, 2),
            "host_syncs_per_step": round(wire.get("host_syncs", 0) / nst, 2),
            "protocol_counts_note": "collectives: all-gathers of meta / bundle lists / exchange table + the two all-to-alls (rows, CIGAR words); the junction "
                                    "counts ride in the next step's first gather.  host syncs: read-backs of the protocol itself (the gathered table; the "
                                    "owner's group count) — tiecov's own read-back is the plain path's too",
            "dist_mode": args.dist_mode,
            "dist_phase_host_ms_per_step": ph,
            "dist_note": "rank 0's figures.  plain_ms_per_step: the plain single-GPU step (collapse + chain + tiecov, %d contexts) on this "
                         "rank's own tile, timed after the timed region; shuffle_ms_per_step: host wall time of the main thread between the "
                         "local collapse and the owner's reduce (cut search, pack, exchange, unpack) — it overlaps the worker thread's next "
                         "local collapse; wire bytes: rows x 40 B + CIGAR words, all destinations (off_rank: without the self block)" % NCTX,
        }

    # ---- R virtual ranks on this one GPU (--force-dist --emulate-world R): every virtual rank holds the per-rank workload (its own files of
    # the synthetic job), the collectives are served in process.  Not a timing of R GPUs — the ranks run one after the other — but the
    # R-way protocol itself: do the lists settle every cut, how many collectives and read-backs a step takes, what the stages cost.
    if use_dist and world == 1 and args.emulate_world > 1 and args.dist_mode == "partials":
        R = args.emulate_world
        ectx = api.Context(local_rank)
        ectx.use_torch_stream()
        etiles = [dtile] + [synth_dev.make_tile_device(files, reads, SYNTH_PROFILE[profile], device=dev, first_file=r * files, tx=tx) for r in range(1, R)]
        eopts = ectx.make_opts(**strat)
        efins = [ectx.collapse(t_, opts=eopts, want_coords=True, want_effend=True, out={}, cap_groups=cap_groups) for t_ in etiles]
        torch.cuda.synchronize()
        em = {}
        for mode_ in ("lists", "rounds"):
            sts = [dict() for _ in range(R)]
            for rep_ in range(3):
                sts = [dict() for _ in range(R)]
                torch.cuda.synchronize()
                te = time.perf_counter()
                eres = tdist.run_loopback(ectx, etiles, [r * files for r in range(R)], per_rank=[dict(local=efins[r], stats=sts[r]) for r in range(R)],
                                          want_coverage=False, device_chain=True, cut_search=mode_, **strat)
                torch.cuda.synchronize()
                te = time.perf_counter() - te
            yc_tot = sum(float(r_.yc.sum()) for r_ in eres)
            em[mode_] = {"ms_per_rank_step_serialised": round(te * 1e3 / R, 3), "cut_rounds_max": max(st_.get("cut_rounds", 0) for st_ in sts),
                         "collectives_per_step": sts[0].get("collectives"), "host_syncs_per_step": sts[0].get("host_syncs"),
                         "groups_out": int(sum(r_.n_groups for r_ in eres)), "groups_per_rank": [int(r_.n_groups) for r_ in eres],
                         "count_conserved": bool(abs(yc_tot - sum(int(f_["n_passed"]) for f_ in efins)) < 0.5)}
        dist_extra["emulated_world"] = {"world": R, "records_per_rank": int(n_records), "partials_per_rank": [int(f_["n_groups"]) for f_ in efins], **em,
                                        "note": "R virtual ranks run one after the other on one GPU with in-process collectives (tiebrush_amd.dist.run_loopback): the "
                                                "protocol's decisions at world R, not its speed on R GPUs"}
        del etiles, efins, eres

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
            gg = ctx.collapse(dtile, opts=opts_defer, want_coords=True, out=cbufs, raw=True, cap_groups=cap_groups)
            take("collapse")
            ctx.finish_yd()
            take("collapse")               # the deferred YD stage belongs to tbk_collapse_tile
            view = ctx.groups_to_cov_in(gg)
            take("chain")                  # tiebrush -> tiecov device chain (representatives gathered into tiecov's input view)
            cc = ctx.coverage(view, out=vbufs, raw=True)
            take("coverage")
        cov_tile_us, cov_call_ms = [], []
        for _ in range(max(0, args.cov_prof_reps)):   # further serialised coverage calls on the same view: the spread of cov_tile
            t1 = time.perf_counter()
            ctx.coverage(view, out=vbufs, raw=True)
            cov_call_ms.append((time.perf_counter() - t1) * 1e3)
            kt = ctx.kernel_times()
            if "cov_tile" in kt:
                cov_tile_us.append(kt["cov_tile"][0] / max(kt["cov_tile"][1], 1) * 1e3)
        ctx.set_profiling(False)
        # algorithmic bytes; the profiled steps are rank 0's local collapse + coverage
        b_collapse = gg["n_passed"] * 16 + 4 * n_cig_in
        ncig_cov = int(view.n_cigar_ops)
        b_cov = gg["n_groups"] * 12 + 4 * ncig_cov + 16 * cc["span_bases"] + 16 * cc["n_intervals"]
        traffic = {}
        tname = "traffic_%s_%dx%d.json" % (profile, files, reads)
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):          # PMC passes (FETCH_SIZE / WRITE_SIZE) of this same workload, per launch: rocprofv3 counters
            traffic = json.load(open(tpath)).get("bytes_per_launch", {})   # cannot be read from inside the process they profile

        def roofline(stage, name, alg_bytes, launch_us=None, owed_bytes=None, owed_note=None):
            ms, ln = acc[(stage, name)]
            per_launch_ms = ms / ln
            launches_per_step = ln / args.prof_steps
            r = {"kernel": name, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            if launch_us:                  # a list of per-launch durations: quote the median, state the spread
                s_ = sorted(launch_us)
                med = s_[len(s_) // 2]
                per_launch_ms = med / 1e3
                r.update(avg_launch_us=round(med, 2), launch_us_min=round(s_[0], 2), launch_us_max=round(s_[-1], 2), launches_measured=len(s_),
                         statistic="median")
            else:
                r.update(avg_launch_us=round(per_launch_ms * 1e3, 2))
            achieved = (alg_bytes / launches_per_step) / (per_launch_ms * 1e-3) / 1e9
            tb = traffic.get(name)
            r.update(achieved=round(achieved, 1), frac=round(achieved / HBM_PEAK_GBS, 4), traffic=tb,
                     frac_traffic=round(tb / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tb else None,
                     traffic_source=("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, per launch)" % tname) if tb else None,
                     launches_per_step=launches_per_step, algorithmic_bytes_per_step=int(alg_bytes),
                     measured="HIP events on the launch stream, profiling steps after the timed region, calls serialised (kernel alone on the GPU)")
            if owed_bytes is not None:     # the bytes this kernel itself has to move in this run (the SURVEY §8d figure counts more)
                r.update(frac_own_bytes=round((owed_bytes / launches_per_step) / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         own_bytes_per_step=int(owed_bytes), own_bytes_note=owed_note)
            return r

        tot = {k: v[0] / args.prof_steps for k, v in acc.items()}
        dom = max(tot, key=tot.get)
        # what the kernels owe in THIS run: the collapse writes no per-record group id (rec_group is not requested: 4 B per record of the
        # §8d figure are never written); cov_tile_k reads the view's records and writes change points — the 16 S bytes of the reference's
        # depth array never exist in HBM, the run-length pass and the interval writes belong to cov_iv_*
        # ... and the window kernels read the NH and MAPQ columns only when -N / -Q refer to them (5 of a record's 12 input bytes)
        opts_ = WORKLOADS[profile][2]
        cols_skipped = 0 if ("max_nh" in opts_ or "min_qual" in opts_) else 5
        b_collapse_owed = gg["n_passed"] * (12 - cols_skipped) + 4 * n_cig_in
        b_cov_own = gg["n_groups"] * 12 + 4 * ncig_cov + 16 * cc["n_intervals"]
        note_c = ("records in (%d B + CIGAR words%s); no group-id write: rec_group is not requested"
                  % (12 - cols_skipped, "; NH and MAPQ are not read: no filter refers to them" if cols_skipped else ""))
        note_v = "view records in (12 B + CIGAR words) + change points out (16 B each, at least one per interval: a lower bound)"
        roof["roofline"] = roofline(dom[0], dom[1], b_collapse if dom[0] == "collapse" else b_cov,
                                    owed_bytes=b_collapse_owed if dom[0] == "collapse" else b_cov_own, owed_note=note_c if dom[0] == "collapse" else note_v)
        if ("coverage", "cov_tile") in acc:
            roof["roofline_coverage"] = roofline("coverage", "cov_tile", b_cov, cov_tile_us or None, owed_bytes=b_cov_own, owed_note=note_v)
            if cov_call_ms:
                s_ = sorted(cov_call_ms)
                roof["roofline_coverage"]["coverage_call_ms_median"] = round(s_[len(s_) // 2], 3)
                roof["roofline_coverage"]["frac_whole_call"] = round(b_cov / (s_[len(s_) // 2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        cdom = max((k for k in tot if k[0] == "collapse"), key=tot.get)
        roof["roofline_collapse"] = roofline("collapse", cdom[1], b_collapse, owed_bytes=b_collapse_owed, owed_note=note_c)
        roof["kernel_ms_per_step"] = {"%s/%s" % k: round(v, 4) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
        roof["launches_per_step"] = int(sum(v[1] for v in acc.values()) / args.prof_steps)
        roof["gpu_kernel_ms_per_step_total"] = round(sum(tot.values()), 4)
        roof["step_frac_of_hbm_peak_algorithmic"] = round((b_collapse + b_cov) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if not use_dist else None

    # ---- kernel path, pinned host -> pinned host: H2D of the input and D2H of every result inside the clock ----
    # The host hands the tile over the way the streaming reader cuts it (TInputFiles::next_tile: sub-tiles bounded by global bundle
    # boundaries, nothing the collapse or tiecov computes crosses one), every sub-tile in the packed wire form (tbk_packed_in: 9
    # bytes per record + the CIGAR words) in pinned memory.  Two contexts, a host thread each, take the sub-tiles in turn: the
    # link carries sub-tile i + 1 in while sub-tile i is collapsed and covered and sub-tile i - 1's results go out.
    host_path = None
    if rank == 0 and not use_dist and not args.no_host_path:
        host_path = host_to_host_leg(args, torch, np, api, dtile, strat, last, n_records, n_cig_in)

    # ---- CPU baseline: the oracle (literal single-threaded restatement of the reference) ----
    cpu = None
    if rank == 0 and not use_dist and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, profile, files, dtile, ctx, strat, n_records)

    if rank == 0:
        line = {
            "metric": "input alignment records/sec collapsed (tiebrush) + bases/sec covered (tiecov)",
            "value": round(tot_records * args.steps / dt, 1),
            "unit": "records/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": ("%s: one job of %d synthetic sorted BAMs x %d 100bp reads, %d files per GPU, %s collapse + tiecov -c -j of the result"
                                    % (profile, job_files, reads, files, strat_name)) if scaling == "strong" else
                                   ("%s: %d synthetic sorted BAMs x %d 100bp reads per GPU, %s collapse + tiecov -c -j of the result"
                                    % (profile, files, reads, strat_name)),
                       "records_per_gpu": int(n_records), "groups_out": int(n_groups), "parallelism": "files-per-rank x%d" % world,
                       "ranks_seen": ranks_seen,
                       "resident": "SoA in HBM before the timed region", "generated_on_device_s": round(t_gen, 2),
                       "contexts": NCTX if not use_dist else 3, "hbm_in_use_gb": hbm_used_gb},
            "bases_per_s": round(tot_bases * args.steps / dt, 1),
            "tiecov": {"bases_covered_per_step": int(n_bases), "bundle_span_bases": int(span), "intervals": int(n_iv), "junctions": int(n_j)},
        }
        line.update(dist_extra)
        if scaling == "strong" and world > 1:
            # the N = 1 end of this fixed job, recorded on an MI355X (profiles/: a run of `--gpus 1 --profile c4 --scaling strong`): the
            # default N = 1 line is config 3, another workload, so a scaling efficiency has to be read against THIS figure
            refs = [os.path.join(ROOT, "profiles", "r%d_bench_%s_strong_n1.json" % (r_, profile)) for r_ in (6, 5, 4)]   # (the newest recorded)
            ref = next((r_ for r_ in refs if os.path.exists(r_)), refs[-1])
            if os.path.exists(ref) and job_files == STRONG_JOB_FILES[profile] and reads == WORKLOADS[profile][1]:
                try:
                    r1 = json.loads(open(ref).read().strip().splitlines()[-1])
                    line["strong_scaling_n1_reference"] = {"value": r1["value"], "ms_per_step": r1["ms_per_step"], "workload": r1["config"]["workload"],
                                                           "source": "profiles/%s (recorded, not measured in this run)" % os.path.basename(ref),
                                                           "speedup_vs_n1": round(line["value"] / r1["value"], 3)}
                except Exception:
                    pass
        line.update(roof)
        # SURVEY.md §8d names two kernel-path figures; `value` is the first (the bench contract: inputs resident in HBM), the second rides
        # beside it at the top level, with the link's own roofline
        line["value_resident"] = line["value"]
        line["value_definitions"] = {"value": "= value_resident: records/s with the SoA tile in HBM before the timed region (the bench contract's `value`)",
                                     "value_host_to_host": "records/s pinned host -> pinned host: H2D of the packed tile, collapse, tiecov, D2H of every result "
                                                           "inside the clock — the figure SURVEY.md §8(d) defines as kernel-path (SURVEY.md:429)",
                                     "end_to_end*": "BAM files -> BAM file through the `tiebrush` command line, process start to exit"}
        if host_path is not None:
            line["value_host_to_host"] = host_path["value"]
            lm = host_path.get("link_measured", {})
            moved = host_path["h2d_bytes"] + host_path["d2h_bytes"]
            ach = moved / (host_path["ms_per_step"] * 1e-3) / 1e9
            # what bounds this leg is the link it crosses twice: bytes on the link / time, against the link's measured two-way rate on this box
            # (the best of what the probe saw: one direction alone or both at once — on these boxes the two directions share one rate)
            pks = [v for v in (lm.get("both_ways_gb_per_s"), lm.get("h2d_gb_per_s"), lm.get("d2h_gb_per_s")) if v]
            pk = max(pks) if pks else None
            line["roofline_link"] = {"bound": "pcie", "achieved": round(ach, 1), "peak": pk, "unit": "GB/s", "frac": round(ach / pk, 4) if pk else None,
                                     "bytes_per_step": int(moved), "ms_per_step": host_path["ms_per_step"],
                                     "h2d_alone_gb_per_s": lm.get("h2d_gb_per_s"), "d2h_alone_gb_per_s": lm.get("d2h_gb_per_s"),
                                     "both_ways_gb_per_s": lm.get("both_ways_gb_per_s"),
                                     "peak_source": "measured in this run: 1 GiB of pinned memory up alone, down alone and each way on two streams at once "
                                                    "(the best of five each; the largest of the three)"}
            line["kernel_path_host_to_host"] = host_path
        if e2e is not None:
            for sub, name in (("seq", "end_to_end_seq"), ("seq_long", "end_to_end_seq_long"), ("c3_options", "end_to_end_c3_options")):
                if isinstance(e2e, dict) and sub in e2e:
                    line[name] = e2e.pop(sub)
            line["end_to_end"] = e2e
        if cpu is not None:
            line["cpu_baseline"] = cpu
        # the files -> files CPU path, timed by the end-to-end leg on that leg's own files (tools/e2e_leg.py: cpu_end_to_end)
        ecpu = line.get("end_to_end_seq", {}).pop("cpu_baseline", None) if isinstance(line.get("end_to_end_seq"), dict) else None
        if ecpu is not None:
            line.setdefault("cpu_baseline", {})["end_to_end"] = dict(ecpu, beside="end_to_end_seq (the same input files, process start to output file)")
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


def host_to_host_leg(args, torch, np, api, dtile, strat, last, n_records, n_cig_in):
    import threading
    from dataclasses import replace

    from tiebrush_amd.soa import PackedTile
    K = max(1, args.host_subtiles)
    dev = dtile.tid.device
    # -- sub-tile cuts at bundle boundaries of the collapsed groups (every record of config 3 passes the filters): from the resident
    #    run's own output — reference id and start from g_key, ends from g_end.  Outside the clock: the reader cuts while it decodes.
    g0 = next(x for x in last if x is not None)[0]
    ng = g0["n_groups"]
    bufs = g0["_bufs"]
    k0 = bufs["g_key"][:2 * ng].view(-1, 2)[:, 0]
    gtid = (k0 >> 33) & 0x7FFFFFFF                                  # tid + 1
    gstart = (k0 >> 2) & 0x7FFFFFFF                                 # 1-based start
    skey = (gtid << 32) | gstart
    ekey = (gtid << 32) | bufs["g_end"][:ng].to(torch.int64)
    reach = torch.cummax(ekey, 0).values
    isb = torch.ones(ng, dtype=torch.bool, device=dev)
    isb[1:] = skey[1:] > reach[:-1]                                 # a group that starts beyond every earlier end opens a bundle
    bidx = torch.nonzero(isb).view(-1)
    want = (torch.arange(1, K, device=dev, dtype=torch.int64) * ng) // K
    pick = bidx[torch.searchsorted(bidx, want).clamp(max=bidx.numel() - 1)] if K > 1 else want
    cut_keys = [int(x) for x in torch.unique(skey[pick]).tolist()] if K > 1 else []   # (tid + 1) << 32 | 1-based start
    del skey, ekey, reach, isb, bidx
    rkey = ((dtile.tid.to(torch.int64) + 1) << 32) | (dtile.pos.to(torch.int64) + 1)
    fo = np.asarray(dtile.file_off, np.int64)
    nf = dtile.n_files
    ck = torch.tensor(cut_keys, dtype=torch.int64, device=dev)
    bounds = np.zeros((nf, len(cut_keys) + 2), np.int64)            # per file: record index of every cut
    for f in range(nf):
        bounds[f, 0], bounds[f, -1] = fo[f], fo[f + 1]
        if len(cut_keys):
            bounds[f, 1:-1] = fo[f] + torch.searchsorted(rkey[int(fo[f]):int(fo[f + 1])], ck).cpu().numpy()
    del rkey
    co = dtile.cig_off.to(torch.int64) & 0xFFFFFFFF
    pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t)
    subs, in_bytes = [], 0
    for j in range(bounds.shape[1] - 1):                            # the packed form of sub-tile j, laid out on the device, then pinned
        rng = [(int(bounds[f, j]), int(bounds[f, j + 1])) for f in range(nf)]
        cat = lambda t: torch.cat([t[a:b] for a, b in rng])
        sfo = np.concatenate([[0], np.cumsum([b - a for a, b in rng])]).astype(np.uint32)
        tid = cat(dtile.tid)
        n = int(tid.numel())
        if n == 0:
            continue
        st = cat(dtile.strand).to(torch.int64)
        sc = torch.where(st == 43, 0, torch.where(st == 45, 1, 2))
        nh = cat(dtile.nh).to(torch.int64)
        assert int(nh.min()) >= 0 and int(nh.max()) <= 1021         # (the synthetic profiles: every NH in range, none absent)
        meta = ((cat(dtile.flag).to(torch.int64) & 0xFFF) | (sc << 12) | (cat(dtile.mapq).to(torch.int64) << 14) | (nh << 22)).to(torch.int32)
        ncig = torch.cat([co[a + 1:b + 1] - co[a:b] for a, b in rng])
        assert int(ncig.max()) < 255
        cig = torch.cat([dtile.cig[int(co[a]):int(co[b])] for a, b in rng])
        brk = torch.ones(n, dtype=torch.bool, device=dev)
        brk[1:] = tid[1:] != tid[:-1]
        brk[torch.from_numpy(sfo[:-1][sfo[:-1] < n].astype(np.int64)).to(dev)] = True
        starts = torch.nonzero(brk).view(-1)
        pt = PackedTile(n_files=nf, file_off=sfo, tid_run_end=torch.cat([starts[1:], torch.tensor([n], device=dev)]).cpu().numpy().astype(np.uint32),
                        tid_run_tid=tid[starts].cpu().numpy().astype(np.int32), pos=pin(cat(dtile.pos)), meta=pin(meta), ncig=pin(ncig.to(torch.uint8)),
                        cig=pin(cig), nh_esc_idx=np.zeros(0, np.uint32), nh_esc_val=np.zeros(0, np.int32), ncig_esc_idx=np.zeros(0, np.uint32),
                        ncig_esc_val=np.zeros(0, np.uint32))
        in_bytes += sum(int(t.numel()) * t.element_size() for t in (pt.pos, pt.meta, pt.ncig, pt.cig)) + pt.tid_run_end.nbytes * 2
        subs.append(pt)
        del tid, st, sc, nh, meta, ncig, cig, brk
    torch.cuda.synchronize()
    # contexts of the leg's own (arenas sized by a sub-tile, not by the tile): one more than two keeps a sub-tile ready for the link
    # whenever it comes free
    NC = max(1, args.host_contexts)
    ctxs = [api.Context(dev.index) for _ in range(NC)]
    opts_defer = ctxs[0].make_opts(defer_yd=True, **strat)
    cb2, vb2, hout = [{} for _ in range(NC)], [{} for _ in range(NC)], [dict() for _ in range(NC)]
    capg = max(1 << 20, max(p.n_records for p in subs) // 4)
    totals = {"passed": 0, "groups": 0, "iv": 0, "j": 0, "out_bytes": 0, "t_h2d": 0.0, "t_compute": 0.0, "t_d2h": 0.0}
    lock = threading.Lock()
    link = threading.Semaphore(int(os.environ.get("TBK_H2H_LINK", "1")))

    def run_all():
        errs = []
        for kk in totals:
            totals[kk] = 0

        def worker(i):
            torch.cuda.set_device(dev)
            cx = ctxs[i]
            try:
                for j in range(i, len(subs), NC):
                    ta = time.perf_counter()
                    with link:                                      # one sub-tile on the link at a time, at its full rate: the contexts fall
                        t = cx.unpack_tile(subs[j])                 # out of step (two copies side by side finish together and leave the link
                    tb_ = time.perf_counter()                       # idle while both compute); H2D of the packed arrays + expansion
                    g = cx.collapse(t, opts=opts_defer, want_coords=True, out=cb2[i], raw=True, cap_groups=capg)
                    view = cx.groups_to_cov_in(g)
                    c = cx.coverage(view, out=vb2[i], raw=True)
                    cx.finish_yd()
                    tc_ = time.perf_counter()
                    ng_, ni, nj = g["n_groups"], c["n_intervals"], c["n_junctions"]
                    outs = [(cb2[i][k_], ng_) for k_ in ("rep", "yc", "yx", "yd", "g_start", "g_end")]
                    outs += [(vb2[i][k_], ni) for k_ in ("iv_tid", "iv_start", "iv_end", "iv_val")]
                    outs += [(vb2[i][k_], nj) for k_ in ("j_tid", "j_start", "j_end", "j_strand", "j_val")]
                    nb = 0
                    for q, (tt, cnt) in enumerate(outs):
                        if q not in hout[i] or hout[i][q].numel() < cnt:
                            hout[i][q] = torch.empty(max(int(cnt * 1.2), 1), dtype=tt.dtype, pin_memory=True)
                        hout[i][q][:cnt].copy_(tt[:cnt], non_blocking=True)
                        nb += cnt * tt.element_size()
                    torch.cuda.current_stream().synchronize()       # (this sub-tile's results are in host memory)
                    td_ = time.perf_counter()
                    with lock:
                        totals["t_h2d"] += tb_ - ta
                        totals["t_compute"] += tc_ - tb_
                        totals["t_d2h"] += td_ - tc_
                        totals["passed"] += g["n_passed"]
                        totals["groups"] += ng_
                        totals["iv"] += ni
                        totals["j"] += nj
                        totals["out_bytes"] += nb
            except BaseException as e:
                errs.append(e)

        th = [threading.Thread(target=worker, args=(i,)) for i in range(NC)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if errs:
            raise errs[0]

    run_all()                                                       # warm-up: arenas, unpack buffers, pinned result buffers
    ref_g, ref_c = next(x for x in last if x is not None)
    assert totals["passed"] == ref_g["n_passed"] and totals["groups"] == ref_g["n_groups"], "sub-tiles disagree with the one-tile run"
    assert totals["iv"] == ref_c["n_intervals"] and totals["j"] == ref_c["n_junctions"], "sub-tiles disagree with the one-tile run (tiecov)"
    reps = 3
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        run_all()
    torch.cuda.synchronize()
    hdt = (time.perf_counter() - t1) / reps
    for cx in ctxs:
        cx.close()
    soa_bytes = n_records * 20 + 4 + 4 * n_cig_in
    # the link's own rate on this box, measured here: 1 GiB of pinned memory each way, alone and both ways at once (two streams)
    link = {}
    try:
        nb_ = 1 << 30
        hp_a = torch.empty(nb_, dtype=torch.uint8, pin_memory=True)
        hp_b = torch.empty(nb_, dtype=torch.uint8, pin_memory=True)
        dv_a = torch.empty(nb_, dtype=torch.uint8, device=dev)
        dv_b = torch.empty(nb_, dtype=torch.uint8, device=dev)
        s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

        def timed(fn, reps_=5):        # (the best of five: what the link can do is the ceiling the leg is priced against)
            fn()
            torch.cuda.synchronize()
            best = None
            for _ in range(reps_):
                t_ = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - t_
                best = dt_ if best is None else min(best, dt_)
            return best

        def up():
            with torch.cuda.stream(s_a):
                dv_a.copy_(hp_a, non_blocking=True)

        def down():
            with torch.cuda.stream(s_b):
                hp_b.copy_(dv_b, non_blocking=True)

        def both():
            up()
            down()

        link = {"h2d_gb_per_s": round(nb_ / timed(up) / 1e9, 1), "d2h_gb_per_s": round(nb_ / timed(down) / 1e9, 1),
                "both_ways_gb_per_s": round(2 * nb_ / timed(both) / 1e9, 1)}
        del hp_a, hp_b, dv_a, dv_b
    except Exception as e:
        link = {"error": repr(e)}
    return {"value": round(totals["passed"] / hdt, 1), "link_measured": link, "unit": "records/s", "ms_per_step": round(hdt * 1e3, 3), "h2d_bytes": int(in_bytes),
            "d2h_bytes": int(totals["out_bytes"]), "reps": reps, "sub_tiles": len(subs), "contexts": NC,
            "h2d_bytes_as_soa": int(soa_bytes), "link_gb_per_s": round((in_bytes + totals["out_bytes"]) / hdt / 1e9, 1),
            "host_wall_ms_summed_over_sub_tiles": {"unpack (H2D + expansion)": round(totals["t_h2d"] * 1e3, 1),
                                                   "collapse + tiecov + YD": round(totals["t_compute"] * 1e3, 1),
                                                   "results D2H": round(totals["t_d2h"] * 1e3, 1)},
            "note": "the tile as %d sub-tiles cut at bundle boundaries (what the streaming reader hands over), each in the packed wire form "
                    "(tbk_packed_in: 9 B per record + CIGAR words) in pinned host memory -> groups, intervals and junctions of every sub-tile in "
                    "pinned host memory; H2D, expansion, collapse, tiecov and D2H inside the clock, two contexts overlapping transfer and compute; "
                    "totals checked against the one-tile run" % len(subs)}


def cpu_baseline(args, profile, files, dtile, ctx, strat, n_records):
    """1 thread at -O2 (how one would build it) and at -O0 (how the reference ships, CMakeLists.txt:49), and the reference's own
    best-case parallel mode: tiewrap-style batches (tiewrap.py:96-126: `-t cores -b ceil(k / cores)`, every batch collapsed by its own
    process, the batch outputs collapsed again as TieBrush-merged inputs)."""
    import multiprocessing as mp
    import shutil
    import tempfile

    import numpy as np

    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth, synth_dev
    from tiebrush_amd.soa import SoATile
    okw = ORACLE_KW[profile]
    import torch
    dev = dtile.tid.device
    tile_bytes = sum(int(getattr(dtile, k).numel()) * getattr(dtile, k).element_size() for k in ("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig"))
    budget = host_memory_budget()
    want = args.cpu_sample_records or n_records
    if want >= n_records and tile_bytes * 3.2 + (8 << 30) > budget:      # host copy + its /dev/shm batches + outputs
        want = int(n_records * max(0.02, (budget - (8 << 30)) / (tile_bytes * 3.2)))
    if want >= n_records:
        sample = synth_dev.tile_to_host(dtile)
        wdesc = "the whole tile"
    else:
        n_t2 = int((dtile.tid == 2).sum())
        w = int(synth.REF_LENS[2] * min(1.0, want / max(n_t2, 1)))
        sample = synth_dev.tile_to_host(dtile, window=(2, 0, w))
        wdesc = "all %d files restricted to chr3:0-%d" % (files, w)

    orc.lib("O2"), orc.lib("O0")                                          # (loading the libraries: before the clock)
    t1 = time.perf_counter()
    og = orc.collapse(sample, opt="O2", **okw)
    oc = orc.coverage(synth.collapsed_to_cov_input(sample, og), opt="O2")
    d2 = time.perf_counter() - t1
    v2 = og["n_passed"] / d2
    # the -O0 leg on a bounded slice of the sample (about 10 s of work): the ratio is what matters
    n0 = min(sample.n_records, int(max(1.0, v2 / 2.0) * 10.0))
    if n0 >= sample.n_records:
        s0 = sample
    else:
        frac = n0 / sample.n_records
        fo = sample.file_off.astype(np.int64)
        key = (sample.tid.astype(np.int64) << 32) | sample.pos.astype(np.int64)
        kcut = np.sort(key[::max(1, len(key) // 4096)])[int(frac * min(4096, len(key) - 1))]
        sel = [(int(fo[f]), int(fo[f]) + int(np.searchsorted(key[fo[f]:fo[f + 1]], kcut))) for f in range(sample.n_files)]
        co = sample.cig_off.astype(np.int64)
        cat = lambda a: np.concatenate([a[x:y] for x, y in sel])
        nfo = np.zeros(sample.n_files + 1, np.uint32)
        nfo[1:] = np.cumsum([y - x for x, y in sel])
        ncg = np.concatenate([co[x + 1:y + 1] - co[x:y] for x, y in sel])
        s0 = SoATile(n_files=sample.n_files, file_off=nfo, tbmerged=sample.tbmerged.copy(), tid=cat(sample.tid), pos=cat(sample.pos),
                     flag=cat(sample.flag), mapq=cat(sample.mapq), strand=cat(sample.strand), nh=cat(sample.nh),
                     cig_off=np.concatenate([[0], np.cumsum(ncg)]).astype(np.uint32),
                     cig=np.concatenate([sample.cig[int(co[x]):int(co[y])] for x, y in sel]))
    t1 = time.perf_counter()
    o0 = orc.collapse(s0, opt="O0", **okw)
    orc.coverage(synth.collapsed_to_cov_input(s0, o0), opt="O0")
    v0 = o0["n_passed"] / (time.perf_counter() - t1)
    # the GPU path on the same sample must agree with the oracle (counts here; tests/ compare every array)
    if sample.n_records <= 64_000_000:
        sg = ctx.collapse(api.to_device(sample, str(dev)), **strat)
        assert sg["n_passed"] == og["n_passed"] and sg["n_groups"] == og["n_groups"], "GPU/oracle disagree on the bench sample"
    cpu = {"value": round(v2, 1), "unit": "records/s", "cores": 1, "kind": "port",
           "sample": "%s: %d records -> %d groups, collapse + coverage on SoA, gcc -O2, 1 pass (%.1f s)" % (wdesc, sample.n_records, og["n_groups"], d2),
           "value_O0": round(v0, 1), "note_O0": "same code at -O0 -g, how the reference ships (CMakeLists.txt:49), on the first %d records of the sample" % s0.n_records,
           "host_cores_available": os.cpu_count(), "host_cpu_quota": cpu_budget()}

    # ---- tiewrap-style parallel line ----
    procs = args.cpu_procs or min(16, cpu_budget(), sample.n_files)
    if procs >= 2:
        shm = tempfile.mkdtemp(prefix="tbk_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            k = sample.n_files
            bsz = -(-k // procs)                           # tiewrap: -b ceil(k / cores)
            nb = -(-k // bsz)
            fo, co = sample.file_off.astype(np.int64), sample.cig_off.astype(np.int64)
            for b in range(nb):                            # (input staging: outside the clock, like reading the BAMs would be)
                f0, f1 = b * bsz, min(k, (b + 1) * bsz)
                lo, hi = int(fo[f0]), int(fo[f1])
                np.save(os.path.join(shm, "b%d_file_off.npy" % b), (fo[f0:f1 + 1] - fo[f0]).astype(np.uint32))
                for nm in ("tid", "pos", "flag", "mapq", "strand", "nh"):
                    np.save(os.path.join(shm, "b%d_%s.npy" % (b, nm)), getattr(sample, nm)[lo:hi])
                np.save(os.path.join(shm, "b%d_cig_off.npy" % b), (co[lo:hi + 1] - co[lo]).astype(np.uint32))
                np.save(os.path.join(shm, "b%d_cig.npy" % b), sample.cig[int(co[lo]):int(co[hi])])
            mpx = mp.get_context("spawn")                  # (never fork a process that holds a GPU context)
            with mpx.Pool(min(procs, nb)) as pool:
                pool.map(_cpu_warm, range(4 * procs))          # (interpreter start and imports: before the clock)
                t1 = time.perf_counter()
                outs = pool.map(_cpu_batch_worker, [(shm, b, okw) for b in range(nb)])
                t_batches = time.perf_counter() - t1
            # second level: the batch outputs as TieBrush-merged inputs of one more run (tiewrap.py:120-126)
            t1 = time.perf_counter()
            ld = lambda b, nm: np.load(os.path.join(shm, "o%d_%s.npy" % (b, nm)))
            ngs = [o[2] for o in sorted(outs)]
            fo2 = np.concatenate([[0], np.cumsum(ngs)]).astype(np.uint32)
            ncs = [ld(b, "cig_off") for b in range(nb)]
            co2 = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(c.astype(np.int64)) for c in ncs]))]).astype(np.uint32)
            cat = lambda nm: np.concatenate([ld(b, nm) for b in range(nb)])
            t2 = SoATile(n_files=nb, file_off=fo2, tbmerged=np.ones(nb, np.uint8), tid=cat("tid"), pos=cat("pos"), flag=cat("flag"),
                         mapq=cat("mapq"), strand=cat("strand"), nh=cat("nh"), cig_off=co2, cig=cat("cig"), yc_in=cat("yc"), yx_in=cat("yx"),
                         yd_in=cat("yd"))
            og2 = orc.collapse(t2, opt="O2", **okw)
            orc.coverage(synth.collapsed_to_cov_input(t2, og2), opt="O2")
            t_final = time.perf_counter() - t1
            assert og2["n_groups"] == og["n_groups"] and float(og2["yc"].sum()) == float(og["yc"].sum()), "hierarchical CPU run disagrees with the flat one"
            cpu["parallel"] = {"value": round(og["n_passed"] / (t_batches + t_final), 1), "unit": "records/s", "cores": min(procs, nb),
                               "kind": "port", "mode": "tiewrap-style: %d batches of %d files, one process each (%.1f s), then one run over the %d "
                                                       "batch outputs as TieBrush-merged inputs + coverage (%.1f s); tiewrap.py:96-126" %
                                                       (nb, bsz, t_batches, nb, t_final),
                               "sample": "the same sample; batch inputs staged in /dev/shm outside the clock"}
        finally:
            shutil.rmtree(shm, ignore_errors=True)
    return cpu


if __name__ == "__main__":
    main()
