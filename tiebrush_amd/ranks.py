"""tiebrush over several GPUs of one node, ending in ONE collapsed BAM (`tiebrush --ranks N ...`, or this module under torchrun).

The reference's only multi-worker scheme is tiewrap.py:96-126 — batches of files collapsed by separate processes, their outputs
collapsed again — and it ends in one BAM.  Here the input files shard by rank (one process per GPU, torch.distributed: RCCL over
xGMI), every rank decodes and collapses its own files, group partials are exchanged and reduced by the owner of a coordinate
range (tiebrush_amd/dist.py), and then the winners go home:

  1. the owner of a range knows, per output group, the representative as (global file, index in file) plus YC / YX / YD
     (`ShardResult`); it asks the rank that holds the file for the record — one small all-to-all of (file, index) pairs;
  2. that rank fetches the raw BAM records behind the indices (tbk_bam_records: the inflated input is still on its GPU) and ships
     them back — one all-to-all of lengths, one of bytes;
  3. the owner tags its records as flushPData does (tiebrush.cpp:506-525), frames them and deflates them into BGZF members on its
     share of the host cores (libtbh.so: tbh_tag_deflate_part) — a part file per rank;
  4. rank 0 writes the header the single-GPU command line would write (TInputFiles over all inputs), appends the parts in rank
     order and the EOF member (BGZF members concatenate; GSam.h:648-653 is the reference's writer).
The ranges are in coordinate order and whole bundles each, so the concatenation is the flat run's output record for record.

Refused, as in tiebrush_amd.dist: -A and --store-frac (order-dependent sums across ranks), -F, -M.  -L: the representatives' MD strings
travel beside the rows (tbk_partial_pack_md).
"""
from __future__ import annotations

import argparse
import contextlib
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

VERSION = "0.0.7"


def _parse(argv):
    ap = argparse.ArgumentParser(prog="tiebrush --ranks", add_help=True,
                                 description="multi-GPU tiebrush: the input files shard over the ranks, the output is one BAM")
    ap.add_argument("--ranks", type=int, default=0, help="start this many processes (one per GPU); omit under torchrun")
    ap.add_argument("-o", dest="out", required=True)
    ap.add_argument("-P", "--clip", action="store_true")
    ap.add_argument("-E", "--exon", action="store_true")
    ap.add_argument("-L", "--full", action="store_true")
    ap.add_argument("-S", "--keep-supp", action="store_true")
    ap.add_argument("--keep-secondary", action="store_true")
    ap.add_argument("-N", type=int, default=2**31 - 1)
    ap.add_argument("-Q", type=int, default=-1)
    ap.add_argument("-A", "--collapse-same", action="store_true")
    ap.add_argument("--store-frac", action="store_true")
    ap.add_argument("-M", "--keep-unmap", action="store_true")
    ap.add_argument("-F", type=int, default=0)
    ap.add_argument("-V", "--verbose", action="store_true")
    ap.add_argument("--level", type=int, default=6, help="deflate level of the output")
    ap.add_argument("inputs", nargs="+")
    a = ap.parse_args(argv)
    if a.collapse_same or a.store_frac:
        ap.error("-A and --store-frac need the single-GPU path's ordered passes (run tiebrush without --ranks)")
    if a.F or a.keep_unmap:
        ap.error("-F and -M are not supported by the GPU build")
    if sum([a.clip, a.exon, a.full]) > 1:
        ap.error("only one merging strategy can be requested")
    return a


def _input_files(inputs):
    """the reference's input convention (tmerge.cpp:287-310): one argument that is neither BAM nor SAM is a list of paths"""
    if len(inputs) == 1:
        with open(inputs[0], "rb") as f:
            magic = f.read(4)
        if magic[:2] != b"\x1f\x8b" and not magic.startswith(b"@"):
            out = []
            for line in open(inputs[0]):
                s = line.strip()
                if len(s) >= 2 and not s.startswith("#"):
                    out.append(s)
            inputs = out
    return [os.path.realpath(p) for p in inputs]


def _launch(n, argv):
    """start n ranks of this module (before anything touches a GPU) and wait for them"""
    port = 29400 + (os.getpid() % 2000)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "tiebrush_amd.ranks"] + argv, env=env,
                                      cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    # a rank that fails leaves the others waiting in a collective: when one ends with an error the rest are ended too (these exact
    # processes, by handle)
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = max(rc, abs(r))
                for q in live:
                    q.terminate()
    return rc


def _host_lib():
    from . import _lib
    return _lib.load_host()


def _a2a(dist, torch, x, send_cnt, recv_cnt, dev, max_bytes=256 << 20):
    """all_to_all of the leading-dimension blocks of x; blocks beyond max_bytes go out in rounds (see dist.run_distributed)"""
    world = dist.get_world_size()
    out = torch.empty((int(sum(recv_cnt)),) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
    if world == 1:
        out.copy_(x[:out.shape[0]])
        return out
    row = max(1, x.element_size() * int(np.prod(x.shape[1:], dtype=np.int64)))
    chunk = max(1, max_bytes // row)
    rounds = (max(int(max(send_cnt)), int(max(recv_cnt))) + chunk - 1) // chunk
    t = torch.tensor([rounds], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rounds = max(1, int(t))
    so = np.concatenate([[0], np.cumsum(send_cnt)]).astype(np.int64)
    ro = np.concatenate([[0], np.cumsum(recv_cnt)]).astype(np.int64)
    for r in range(rounds):
        ss = [max(0, min(chunk, int(send_cnt[d]) - r * chunk)) for d in range(world)]
        rs = [max(0, min(chunk, int(recv_cnt[s]) - r * chunk)) for s in range(world)]
        xin = torch.cat([x[int(so[d]) + r * chunk: int(so[d]) + r * chunk + ss[d]] for d in range(world)]) if rounds > 1 else x
        tmp = torch.empty((sum(rs),) + tuple(x.shape[1:]), dtype=x.dtype, device=dev) if rounds > 1 else out
        dist.all_to_all_single(tmp, xin.contiguous(), output_split_sizes=rs, input_split_sizes=ss)
        if rounds > 1:
            o = 0
            for s in range(world):
                out[int(ro[s]) + r * chunk: int(ro[s]) + r * chunk + rs[s]] = tmp[o:o + rs[s]]
                o += rs[s]
    return out


class _Compute:
    """tiebrush_amd.dist's compute object over one context"""

    def __init__(self, ctx):
        self.ctx = ctx

    def collapse(self, tile, **kw):
        return self.ctx.collapse(tile, **kw)

    def groups_to_cov_in(self, fin):
        return self.ctx.groups_to_cov_in(fin)

    def __getattr__(self, name):
        if name.startswith("shard_") or name.startswith("partial_"):
            return getattr(self.ctx, name)
        raise AttributeError(name)


def _device_tile(ctx, torch, s, fo, tbmerged, dev):
    """the SoATile (torch tensors) of a tile tbk_bam_decode left on the device: the arrays are copied out of the decoder's buffers
    (device to device), which then only keep the inflated records for tbk_bam_records"""
    from .soa import SoATile
    hip = C.CDLL("libamdhip64.so")
    n, nc = int(s.n_records), int(s.n_cigar_ops)

    def grab(ptr, count, dt):
        t = torch.empty(max(count, 1), dtype=dt, device=dev)
        if count and ptr:
            rc = hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(count * t.element_size()), 3)   # device to device
            if rc != 0:
                raise RuntimeError("hipMemcpy failed (%d)" % rc)
        return t[:count]

    tile = SoATile(n_files=len(fo) - 1, file_off=np.asarray(fo, np.uint32).copy(), tbmerged=np.asarray(tbmerged, np.uint8).copy(),
                   tid=grab(s.tid, n, torch.int32), pos=grab(s.pos, n, torch.int32), flag=grab(s.flag, n, torch.int16),
                   mapq=grab(s.mapq, n, torch.uint8), strand=grab(s.strand, n, torch.uint8), nh=grab(s.nh, n, torch.int32),
                   cig_off=grab(s.cig_off, n + 1, torch.int32), cig=grab(s.cig, nc, torch.int32))
    if np.any(tile.tbmerged):
        tile.yc_in, tile.yx_in, tile.yd_in = grab(s.yc_in, n, torch.float64), grab(s.yx_in, n, torch.int64), grab(s.yd_in, n, torch.int64)
    if s.md_off:        # -L: the MD strings as CSR (tbk_bam_decode with want_md)
        tile.md_off, tile.md_has = grab(s.md_off, n + 1, torch.int32), grab(s.md_has, n, torch.uint8)
        torch.cuda.synchronize()
        tile.md = grab(s.md, int(tile.md_off[n].item()) if n else 0, torch.uint8)
    torch.cuda.synchronize()
    return tile


def worker(a, argv):
    import torch
    import torch.distributed as dist

    from . import api
    from . import dist as tdist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29400")
    # TBK_RANKS_BACKEND=gloo: several ranks may share one GPU (collectives staged through the host) — the 1-GPU test box
    backend = os.environ.get("TBK_RANKS_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if backend == "nccl" and world > ndev:
        raise SystemExit("Error: %d ranks but %d GPUs (one process per GPU)" % (world, ndev))
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else "cpu"
    t_start = time.perf_counter()
    timing = os.environ.get("TBK_TIMING") is not None
    files = _input_files(a.inputs)
    k = len(files)
    if k < world:
        raise SystemExit("Error: %d input files cannot shard over %d ranks" % (k, world))
    bounds = [(k * r) // world for r in range(world + 1)]
    mine = files[bounds[rank]:bounds[rank + 1]]
    H = _host_lib()
    tb = np.array([H.tbh_is_tiebrush(p.encode()) for p in mine], np.int32)
    if np.any(tb < 0):
        raise SystemExit("Error: cannot read the header of %s" % mine[int(np.argmin(tb))])
    strategy = "clip" if a.clip else ("exon" if a.exon else ("full" if a.full else "cigar"))
    filters = dict(max_nh=a.N, min_qual=a.Q, keep_supplementary=a.keep_supp, keep_secondary=a.keep_secondary)
    if a.verbose and rank == 0:
        sys.stderr.write("Running TieBrush %s on %d ranks. Command line:\ntiebrush %s\n" % (VERSION, world, " ".join(argv)))
    # ---- decode this rank's files on its GPU; the inflated records stay there for the write-back ----
    ctx = api.Context(local_rank)
    raw = [open(p, "rb").read() for p in mine]
    s, fo = ctx.bam_decode(raw, tbmerged=tb.astype(np.uint8), want_md=a.full)
    del raw
    tile = _device_tile(ctx, torch, s, fo, tb.astype(np.uint8), dev)
    t_dec = time.perf_counter()
    # ---- collapse locally, exchange group partials, reduce on the owner (dist.partials_collapse) ----
    res = tdist.run_distributed(_Compute(ctx), tile, bounds[rank], device=dev, want_coverage=False, device_chain=True, strategy=strategy,
                                **filters)
    ng = int(res.n_groups)
    t_col = time.perf_counter()
    # ---- write-back 1: ask the file owners for the representatives ----
    fidx = res.rep_fidx.to(torch.int64).cpu().numpy() if ng else np.zeros(0, np.int64)
    ridx = res.rep_idx.to(torch.int64).cpu().numpy() if ng else np.zeros(0, np.int64)
    owner = np.searchsorted(np.asarray(bounds[1:], np.int64), fidx, side="right")
    order = np.argsort(owner, kind="stable")
    send_cnt = np.bincount(owner, minlength=world).astype(np.int64)
    req = np.stack([fidx[order] - np.asarray(bounds, np.int64)[owner[order]], ridx[order]], axis=1) if ng else np.zeros((0, 2), np.int64)
    c = torch.from_numpy(send_cnt).to(cdev)
    rc_ = torch.empty_like(c)
    dist.all_to_all_single(rc_, c)
    recv_cnt = rc_.cpu().numpy().astype(np.int64)
    got = _a2a(dist, torch, torch.from_numpy(np.ascontiguousarray(req)).to(cdev), send_cnt.tolist(), recv_cnt.tolist(), cdev).cpu().numpy()
    # ---- write-back 2: the raw records behind the requests, back to the askers ----
    tix = (np.asarray(fo, np.int64)[got[:, 0]] + got[:, 1]).astype(np.uint32) if len(got) else np.zeros(0, np.uint32)
    blob, off = ctx.bam_records(tix) if len(tix) else (b"", np.zeros(1, np.uint64))
    lens = np.diff(np.asarray(off, np.int64))
    src_off = np.concatenate([[0], np.cumsum(recv_cnt)])
    bytes_to = np.array([int(lens[src_off[s_]:src_off[s_ + 1]].sum()) for s_ in range(world)], np.int64)   # what goes back to each asker
    c = torch.from_numpy(bytes_to).to(cdev)
    rb = torch.empty_like(c)
    dist.all_to_all_single(rb, c)
    bytes_from = rb.cpu().numpy().astype(np.int64)
    my_lens = _a2a(dist, torch, torch.from_numpy(np.ascontiguousarray(lens)).to(cdev), recv_cnt.tolist(), send_cnt.tolist(), cdev).cpu().numpy()
    bl = torch.frombuffer(bytearray(blob), dtype=torch.uint8) if len(blob) else torch.zeros(0, dtype=torch.uint8)
    my_blob = _a2a(dist, torch, bl.to(cdev), bytes_to.tolist(), bytes_from.tolist(), cdev).cpu().numpy()
    ctx.bam_release()
    t_wb = time.perf_counter()
    # ---- write-back 3: tag, frame, deflate this rank's slice of the output ----
    rec_off_sorted = np.concatenate([[0], np.cumsum(my_lens)])[:-1].astype(np.uint64) + np.uint64(4)   # (skip the block_size field)
    rec_len_sorted = (my_lens - 4).astype(np.uint32)
    rec_off = np.empty(ng, np.uint64)
    rec_len = np.empty(ng, np.uint32)
    rec_off[order] = rec_off_sorted
    rec_len[order] = rec_len_sorted
    yc = np.ascontiguousarray(res.yc.cpu().numpy() if ng else np.zeros(0), np.float64)
    yx = np.ascontiguousarray(res.yx.cpu().numpy() if ng else np.zeros(0), np.int64)
    yd = np.ascontiguousarray(res.yd.cpu().numpy() if ng else np.zeros(0), np.int32)
    part = "%s.part%d" % (a.out, rank)
    threads = max(1, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)) // world)
    my_blob = np.ascontiguousarray(my_blob)
    if H.tbh_tag_deflate_part(my_blob.ctypes.data, rec_off.ctypes.data, rec_len.ctypes.data, ng, yc.ctypes.data, yx.ctypes.data, yd.ctypes.data,
                              a.level, threads, part.encode()) != 0:
        with contextlib.suppress(OSError):
            os.unlink(part)                        # (no half-written part is left behind)
        raise SystemExit("Error: writing %s failed: %s" % (part, H.tbh_last_error().decode()))
    t_part = time.perf_counter()
    tot = torch.tensor([float(res.n_passed_local), float(ng)], dtype=torch.float64, device=cdev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)     # (also the barrier behind the parts)
    # ---- write-back 4: header + parts + EOF ----
    if rank == 0:
        cmd = ["tiebrush"] + list(argv)
        arr = lambda xs: (C.c_char_p * len(xs))(*[x.encode() for x in xs])
        parts = ["%s.part%d" % (a.out, r) for r in range(world)]
        if H.tbh_write_bam_parts(a.out.encode(), VERSION.encode(), len(cmd), arr(cmd), k, arr(files), world, arr(parts), 1) != 0:
            for pth in parts:                      # (the output is incomplete: nothing that looks like a result stays)
                with contextlib.suppress(OSError):
                    os.unlink(pth)
            raise SystemExit("Error: writing %s failed: %s" % (a.out, H.tbh_last_error().decode()))
        n_in, n_out = int(tot[0]), int(tot[1])
        sys.stderr.write("%d input records written as %d (%.2f%% reduction)\n" % (n_in, n_out, 100.0 - (n_out * 100.0) / max(n_in, 1)))
        if timing:
            sys.stderr.write("ranks %d ms: decode %.1f | collapse + exchange + reduce %.1f | fetch representatives %.1f | tag + deflate %.1f | "
                             "header + concatenation %.1f\n" % (world, (t_dec - t_start) * 1e3, (t_col - t_dec) * 1e3, (t_wb - t_col) * 1e3,
                                                                (t_part - t_wb) * 1e3, (time.perf_counter() - t_part) * 1e3))
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = _parse(argv)
    if "RANK" not in os.environ:
        n = a.ranks or 1
        rest, skip = [], False
        for x in argv:                       # the workers get the command line without --ranks
            if skip:
                skip = False
            elif x == "--ranks":
                skip = True
            elif not x.startswith("--ranks="):
                rest.append(x)
        return _launch(n, rest)
    return worker(a, argv)


if __name__ == "__main__":
    sys.exit(main())
