"""Multi-GPU collapse: input files shard per rank, groups that span ranks are stitched by key.

Why this is exact: everything order-dependent is per file — a sample lives on one rank —
or an associative reduction over a group's members:
  YC  = sum of the per-rank partial YC (integer counts: exact in any order)
  YX  = sum of the per-rank YX (each rank's samples are disjoint files)
  YD  = max of the per-rank YD (the list machine of a sample only sees that sample's groups, and their
        order is a function of the group keys, so the owner rank computes its d values alone)
  rep = argmin over the per-rank representatives of (effective end, global file index, index in file)
        — the greedy k-way merge key of tmerge.h:28-50.
`--store-frac` (FP order) and `-A` (needs the global representative's name) are refused here.

Protocol per tile (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" on CPU):
  1. local collapse (tbk_collapse_tile) -> local groups in output order, with rep_effend
  2. all-gather of 64 sampled group keys per rank  -> R-1 target splitters
  3. each target is moved forward to a GLOBAL bundle boundary (no group of any rank spans it) with a
     few all-reduce(max/min) rounds of one scalar — so every rank ends up owning whole tiecov bundles
  4. all-to-all(v) of the partial groups {tid,pos,strand,yx,yd,prio,ncig | yc | CIGAR words}
  5. stitch = tbk_collapse_tile again over the received partials ("files" = source ranks, all marked
     TieBrush-merged so YC/YX/YD are carried; representative by explicit priority)
  6. tiecov on the owned slice; junction numbers are offset by an all-gather of the per-rank counts.

The rank algorithm is a generator that yields collective requests, so the same code runs over
torch.distributed (`run_distributed`) and over an in-process loopback of R virtual ranks (`run_loopback`,
used to exercise R>1 on a single GPU and in CPU tests).  Arrays are numpy (host tiles) or torch CUDA
tensors (tiles resident in HBM: nothing but a few scalars ever visits the host); the index bookkeeping
between the two HIP collapses (searchsorted / cummax / bincount over group keys) is plumbing.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

import numpy as np

from .soa import SoATile, CovInput

N_SAMPLES = 64
KEY_INF = np.iinfo(np.int64).max


# ---- numpy / torch shim (only the handful of index ops the protocol needs) -----------------------------------
def _is_t(a):
    return type(a).__module__.startswith("torch")


def _torch():
    import torch
    return torch


class _NP:
    i64, f64 = np.int64, np.float64

    @staticmethod
    def to_i64(a):
        a = np.asarray(a)
        if a.dtype == np.uint32:
            return a.astype(np.int64)
        return a.astype(np.int64)

    u32_to_i64 = to_i64
    zeros = staticmethod(lambda n, dt=np.int64, like=None: np.zeros(n, dt))
    full = staticmethod(lambda n, v, like=None: np.full(n, v, np.int64))
    arange = staticmethod(lambda n, like=None: np.arange(n, dtype=np.int64))
    cummax = staticmethod(lambda a: np.maximum.accumulate(a) if len(a) else a)
    cumsum = staticmethod(lambda a: np.cumsum(a))
    searchsorted = staticmethod(lambda a, v, right=False: np.searchsorted(a, v, side="right" if right else "left"))
    repeat = staticmethod(lambda a, r: np.repeat(a, r))
    stack = staticmethod(lambda xs, axis=0: np.stack(xs, axis=axis))
    cat = staticmethod(lambda xs: np.concatenate(xs))
    sort = staticmethod(lambda a: np.sort(a))
    item = staticmethod(lambda a: int(a))
    host = staticmethod(lambda a: np.asarray(a))
    where = staticmethod(lambda c, a, b: np.where(c, a, b))

    @staticmethod
    def bincount(x, minlength, weights=None):
        return np.bincount(x, weights=weights, minlength=minlength).astype(np.int64)

    @staticmethod
    def scalar(v, like=None):
        return np.array([v], np.int64)

    @staticmethod
    def as_dtype(a, name):
        return np.asarray(a).astype({"i32": np.int32, "u16": np.uint16, "u8": np.uint8, "u32": np.uint32, "i64": np.int64,
                                     "f64": np.float64, "u64": np.uint64}[name])


class _TT:
    @staticmethod
    def to_i64(a):
        return a.to(_torch().int64)

    @staticmethod
    def u32_to_i64(a):  # uint32 payloads travel as int32 tensors
        return a.to(_torch().int64) & 0xFFFFFFFF

    @staticmethod
    def zeros(n, dt=None, like=None):
        return _torch().zeros(n, dtype=_torch().int64, device=like.device)

    @staticmethod
    def full(n, v, like=None):
        return _torch().full((n,), v, dtype=_torch().int64, device=like.device)

    @staticmethod
    def arange(n, like=None):
        return _torch().arange(n, dtype=_torch().int64, device=like.device)

    cummax = staticmethod(lambda a: _torch().cummax(a, 0).values if a.numel() else a)
    cumsum = staticmethod(lambda a: _torch().cumsum(a, 0))
    searchsorted = staticmethod(lambda a, v, right=False: _torch().searchsorted(a, v, right=right))
    repeat = staticmethod(lambda a, r: _torch().repeat_interleave(a, r))
    stack = staticmethod(lambda xs, axis=0: _torch().stack(xs, dim=axis))
    cat = staticmethod(lambda xs: _torch().cat(xs))
    sort = staticmethod(lambda a: _torch().sort(a).values)
    item = staticmethod(lambda a: int(a.item()))
    host = staticmethod(lambda a: a.cpu().numpy())
    where = staticmethod(lambda c, a, b: _torch().where(c, a, b))

    @staticmethod
    def bincount(x, minlength, weights=None):
        return _torch().bincount(x, weights=weights, minlength=minlength).to(_torch().int64)

    @staticmethod
    def scalar(v, like=None):
        return _torch().tensor([v], dtype=_torch().int64, device=like.device)

    @staticmethod
    def as_dtype(a, name):
        t = _torch()
        return a.to({"i32": t.int32, "u16": t.int16, "u8": t.uint8, "u32": t.int32, "i64": t.int64, "f64": t.float64,
                     "u64": t.int64}[name])


def _xp(a):
    return _TT if _is_t(a) else _NP


@dataclass
class ShardResult:
    """This rank's slice of the global result, in the reference's output order (arrays: numpy or torch)."""
    n_groups: int
    n_passed_local: int          # passing input records of THIS rank's files (sum over ranks = inCounter)
    tid: Any
    start: Any                   # 1-based
    end: Any
    rep_fidx: Any                # global file index of the representative record
    rep_idx: Any                 # its index inside that file
    yc: Any
    yx: Any
    yd: Any
    cov_input: Any = None        # CovInput (host) or DeviceCovView (device): what tiecov reads back for this slice
    coverage: Optional[dict] = None
    junction_offset: int = 0
    n_partials_received: int = 0


def _gather_cigars(X, cig_off, cig, rep):
    co = X.u32_to_i64(cig_off)
    ncig = co[rep + 1] - co[rep]
    n = int(rep.shape[0])
    off = X.cat([X.zeros(1, like=ncig), X.cumsum(ncig)]) if n else X.zeros(1, like=co)
    rec_of = X.repeat(X.arange(n, like=co), ncig)
    within = X.arange(int(X.item(off[-1])) if n else 0, like=co) - off[:-1][rec_of]
    idx = co[rep][rec_of] + within
    return ncig, off, cig[idx]


def _pack_generic(X, local_tile, loc, first_fidx):
    """Exchange layout of the local groups (the layout tbk_pack_partials produces on the device)."""
    like = local_tile.tid
    rep = X.u32_to_i64(loc["rep"])
    fo = X.to_i64(local_tile.file_off) if not _is_t(like) else \
        _torch().from_numpy(np.asarray(local_tile.file_off).astype(np.int64)).to(like.device)
    lf = X.searchsorted(fo, rep, right=True) - 1
    ncig, _, cig = _gather_cigars(X, local_tile.cig_off, local_tile.cig, rep)
    ycb = loc["yc"].view(_torch().int64) if _is_t(like) else np.ascontiguousarray(loc["yc"], np.float64).view(np.int64)
    P = X.stack([X.to_i64(local_tile.tid[rep]), X.to_i64(local_tile.pos[rep]), X.to_i64(local_tile.strand[rep]),
                 X.to_i64(loc["yx"]), X.to_i64(loc["yd"]), X.to_i64(loc["rep_effend"]),
                 ((lf + first_fidx) << 32) | (rep - fo[lf]), ncig, ycb], axis=1)
    return P, cig


def shard_collapse(compute, local_tile: SoATile, first_fidx: int, rank: int, world: int, strategy="cigar",
                   want_coverage=False, device_chain=False, **filters):
    """Generator: yields ("all_gather"|"all_reduce_max"|"all_reduce_min"|"exchange", payload) requests and is sent
    the result; finally returns a ShardResult.  `compute` provides collapse(tile, **kw) / coverage(cin)
    [/ pack_partials / groups_to_cov_in] — tiebrush_amd.api.Context or a wrapper of it."""
    if filters.get("store_frac") or filters.get("collapse_same"):
        raise ValueError("--store-frac and -A need a global second pass: single-GPU only (DESIGN.md §7)")
    X = _xp(local_tile.tid)
    # on the device the local YD list machine is deferred: it overlaps the exchange, the stitch and tiecov, and its column
    # follows the partial rows in a second, small all-to-all (YD of a final group = max over its partials)
    defer = _is_t(local_tile.tid) and hasattr(compute, "finish_yd") and hasattr(compute, "pack_partials")
    loc = compute.collapse(local_tile, strategy=strategy, want_coords=True, want_effend=True,
                           **(dict(defer_yd=True) if defer else {}), **filters)
    ng = int(loc["n_groups"])
    if _is_t(local_tile.tid) and hasattr(compute, "pack_partials"):
        P, cig, emax = compute.pack_partials(loc, first_fidx, int(local_tile.cig.numel()))     # HIP: pack + running max
        if defer:
            P[:, 4] = 0                             # the YD column is not final yet
    else:
        P, cig = _pack_generic(X, local_tile, loc, first_fidx)
        emax = X.cummax(((P[:, 0] + 1) << 32) | X.to_i64(loc["g_end"]))
    key = ((P[:, 0] + 1) << 32) | X.to_i64(loc["g_start"])          # nondecreasing: local output order is bucket order

    # ---- 2. splitter targets from sampled keys -------------------------------------------------------------
    if ng:
        samp = key[(X.arange(N_SAMPLES, like=key) * ng) // N_SAMPLES]
    else:
        samp = X.full(N_SAMPLES, KEY_INF, like=key)
    allsamp = yield ("all_gather", samp)            # [world, N_SAMPLES]
    if world > 1:
        flat = np.sort(X.host(allsamp).reshape(-1))
        flat = flat[flat != KEY_INF]
        tgt = np.array([int(flat[(j * len(flat)) // world]) if len(flat) else KEY_INF for j in range(1, world)], np.int64)
        p = _torch().from_numpy(tgt).to(key.device) if _is_t(key) else tgt
        # ---- 3. move every cut forward to a global bundle boundary (all R-1 cuts refined together) ----------
        for _ in range(100000):
            i = X.searchsorted(key, p)
            m_local = X.full(world - 1, -1, like=key)
            if ng:
                ii = (i - 1).clamp(min=0) if _is_t(i) else np.maximum(i - 1, 0)
                m_local = X.where(i > 0, emax[ii], m_local)
            m = yield ("all_reduce_max", m_local)
            ok = (m < p) | (p == KEY_INF)           # every earlier group of every rank ends before the cut
            if bool(ok.all()):
                break
            nxt = X.full(world - 1, KEY_INF, like=key)
            if ng:
                i2 = X.searchsorted(key, m, right=True)      # first local group starting after m
                jj = i2.clamp(max=ng - 1) if _is_t(i2) else np.minimum(i2, ng - 1)
                nxt = X.where(i2 < ng, key[jj], nxt)
            nxt = X.where(ok, p, nxt)
            p = yield ("all_reduce_min", nxt)
        dest = X.searchsorted(p, key, right=True)
    else:
        dest = X.zeros(ng, like=key)

    # ---- 4. exchange the partial groups ---------------------------------------------------------------------
    # dest is nondecreasing: per-destination row / CIGAR-word counts are differences of prefix positions
    edges = X.searchsorted(dest, X.arange(world + 1, like=key))
    cnt = edges[1:] - edges[:-1]
    csum = X.cat([X.zeros(1, like=key), X.cumsum(P[:, 7])])
    ccnt = csum[edges[1:]] - csum[edges[:-1]]
    rP, rcnt, rcig = yield ("exchange", (P, cnt, cig, ccnt))

    # ---- 5. stitch: second-level collapse over the received partials ---------------------------------------
    n2 = int(rP.shape[0])
    rcnt_h = np.asarray(rcnt, np.int64)
    file_off = np.zeros(world + 1, np.uint32)
    file_off[1:] = np.cumsum(rcnt_h)
    cig_off2 = X.cat([X.zeros(1, like=rP), X.cumsum(rP[:, 7])]) if n2 else X.zeros(1, like=rP)
    col = (lambda c: rP[:, c].contiguous()) if _is_t(rP) else (lambda c: np.ascontiguousarray(rP[:, c]))
    tile2 = SoATile(
        n_files=world, file_off=file_off, tbmerged=np.ones(world, np.uint8), tid=X.as_dtype(col(0), "i32"),
        pos=X.as_dtype(col(1), "i32"), flag=X.as_dtype(X.zeros(n2, like=rP), "u16"),
        mapq=X.as_dtype(X.full(n2, 255, like=rP), "u8"), strand=X.as_dtype(col(2), "u8"),
        nh=X.as_dtype(X.full(n2, -(2**31), like=rP), "i32"), cig_off=X.as_dtype(cig_off2, "u32"), cig=rcig,
        yc_in=col(8).view(_torch().float64) if _is_t(rP) else col(8).view(np.float64), yx_in=col(3), yd_in=col(4),
        prio_hi=col(5) if _is_t(rP) else col(5).view(np.uint64), prio_lo=col(6) if _is_t(rP) else col(6).view(np.uint64))
    fin = compute.collapse(tile2, strategy=strategy, want_coords=True, keep_supplementary=True, keep_secondary=True,
                           want_rec_group=defer)
    g2 = int(fin["n_groups"])
    rep2 = X.u32_to_i64(fin["rep"])
    plo = rP[:, 6][rep2]
    res = ShardResult(n_groups=g2, n_passed_local=int(loc["n_passed"]), tid=tile2.tid[rep2], start=fin["g_start"],
                      end=fin["g_end"], rep_fidx=plo >> 32, rep_idx=plo & 0xFFFFFFFF, yc=fin["yc"], yx=fin["yx"], yd=fin["yd"],
                      n_partials_received=n2)
    # ---- 6. tiecov on the owned slice (whole bundles by construction of the cuts) ---------------------------
    if device_chain:
        res.cov_input = compute.groups_to_cov_in(fin)        # stays in HBM
    else:
        ncg, cof, cg = _gather_cigars(X, tile2.cig_off, tile2.cig, rep2)
        ycf = fin["yc"].to(_torch().float32).to(_torch().float64) if _is_t(rP) else np.asarray(fin["yc"]).astype(np.float32).astype(np.float64)
        res.cov_input = CovInput(tid=tile2.tid[rep2], pos=tile2.pos[rep2], flag=X.as_dtype(X.zeros(g2, like=rP), "u16"),
                                 cig_off=X.as_dtype(cof, "u32"), cig=cg, yc=ycf, strand=tile2.strand[rep2], yx=X.to_i64(fin["yx"]))
    if want_coverage:
        cov = compute.coverage(res.cov_input)
        nj = yield ("all_gather", X.scalar(int(cov["n_junctions"]), like=key))
        res.coverage = cov
        res.junction_offset = int(X.host(nj).reshape(-1)[:rank].sum())
    if defer:
        compute.finish_yd()                                     # local YD column is final now
        ryd, _ = yield ("all_to_all", (X.to_i64(loc["yd"]).contiguous(), cnt))
        yd = _torch().zeros(max(g2, 1), dtype=_torch().int64, device=ryd.device)
        if n2:
            yd.scatter_reduce_(0, fin["rec_group"].to(_torch().int64), ryd, reduce="amax", include_self=True)
        res.yd = yd[:g2].to(_torch().int32)
    return res


# ---- drivers ---------------------------------------------------------------------------------------------------
def run_loopback(compute, tiles, first_fidx, **kw):
    """Run R virtual ranks in one process: steps the R generators in lockstep and serves their collectives."""
    world = len(tiles)
    gens = [shard_collapse(compute, tiles[r], first_fidx[r], r, world, **kw) for r in range(world)]
    reqs = [next(g) for g in gens]
    results = [None] * world
    while any(r is None for r in results):
        assert all(res is None for res in results) and len({q[0] for q in reqs}) == 1, "ranks diverged"
        kind = reqs[0][0]
        pay = [q[1] for q in reqs]
        X = _xp(pay[0][0] if kind in ("exchange", "all_to_all") else pay[0])
        if kind == "all_gather":
            out = [X.stack(pay)] * world
        elif kind == "all_reduce_max":
            out = [X.stack(pay).max(0) if X is _NP else X.stack(pay).max(0).values] * world
        elif kind == "all_reduce_min":
            out = [X.stack(pay).min(0) if X is _NP else X.stack(pay).min(0).values] * world
        elif kind == "all_to_all":
            out = []
            for d in range(world):
                parts, cnts = [], []
                for s_ in range(world):
                    data, cnt = pay[s_]
                    ch = np.asarray(X.host(cnt), np.int64)
                    o = int(ch[:d].sum())
                    parts.append(data[o:o + int(ch[d])])
                    cnts.append(int(ch[d]))
                out.append((X.cat(parts), np.array(cnts, np.int64)))
        elif kind == "exchange":
            out = []
            for d in range(world):
                rows, words, cnts = [], [], []
                for s in range(world):
                    P, cnt, cig, ccnt = pay[s]
                    ch, cc = np.asarray(X.host(cnt), np.int64), np.asarray(X.host(ccnt), np.int64)
                    o, oc = int(ch[:d].sum()), int(cc[:d].sum())
                    rows.append(P[o:o + int(ch[d])])
                    words.append(cig[oc:oc + int(cc[d])])
                    cnts.append(int(ch[d]))
                out.append((X.cat(rows), np.array(cnts, np.int64), X.cat(words)))
        else:
            raise AssertionError(kind)
        new = []
        for r in range(world):
            try:
                new.append(gens[r].send(out[r]))
            except StopIteration as e:
                results[r] = e.value
                new.append(None)
        reqs = new
    return results


def run_distributed(compute, tile, first_fidx, device=None, group=None, **kw):
    """One process per GPU: serve the generator's collectives with torch.distributed (RCCL on ROCm).  Payloads that
    are already torch tensors on the collective's device go out as they are (no host staging)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")

    _signed = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64, np.dtype(np.uint16): np.int16}

    def t(a):
        if _is_t(a):
            return a
        a = np.ascontiguousarray(a)
        if a.dtype in _signed:          # collectives have no unsigned types: ship the same bits as signed
            a = a.view(_signed[a.dtype])
        return torch.from_numpy(a).to(dev)

    def back(x, like):
        if _is_t(like):
            return x
        r = x.cpu().numpy()
        return r.view(like.dtype) if np.asarray(like).dtype in _signed else r

    gen = shard_collapse(compute, tile, first_fidx, rank, world, **kw)
    try:
        req = next(gen)
        while True:
            kind, pay = req
            if kind == "all_gather":
                x = t(pay).contiguous()
                out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
                dist.all_gather_into_tensor(out.view(-1), x.view(-1), group=group)
                res = back(out, pay)
            elif kind in ("all_reduce_max", "all_reduce_min"):
                x = t(pay).clone()
                dist.all_reduce(x, op=dist.ReduceOp.MAX if kind.endswith("max") else dist.ReduceOp.MIN, group=group)
                res = back(x, pay)
            elif kind == "all_to_all":
                data, cnt = pay
                c = t(cnt).contiguous()
                rc = torch.empty_like(c)
                dist.all_to_all_single(rc, c, group=group)
                sc_h, rc_h = c.cpu().numpy().astype(np.int64), rc.cpu().numpy().astype(np.int64)
                x = t(data).contiguous()
                out = torch.empty((int(rc_h.sum()),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                dist.all_to_all_single(out, x, output_split_sizes=rc_h.tolist(), input_split_sizes=sc_h.tolist(), group=group)
                res = (back(out, data), rc_h)
            elif kind == "exchange":
                P, cnt, cig, ccnt = pay
                c = torch.stack([t(cnt), t(ccnt)], dim=1).contiguous()     # [world, 2]: rows and CIGAR words per destination
                rc = torch.empty_like(c)
                dist.all_to_all_single(rc, c, group=group)                  # who sends me how much
                sc_h, rc_h = c.cpu().numpy().astype(np.int64), rc.cpu().numpy().astype(np.int64)
                x = t(P).contiguous()
                outP = torch.empty((int(rc_h[:, 0].sum()),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                dist.all_to_all_single(outP, x, output_split_sizes=rc_h[:, 0].tolist(), input_split_sizes=sc_h[:, 0].tolist(), group=group)
                y = t(cig).contiguous()
                outC = torch.empty(int(rc_h[:, 1].sum()), dtype=y.dtype, device=y.device)
                dist.all_to_all_single(outC, y, output_split_sizes=rc_h[:, 1].tolist(), input_split_sizes=sc_h[:, 1].tolist(), group=group)
                res = (back(outP, P), rc_h[:, 0].copy(), back(outC, cig))
            else:
                raise AssertionError(kind)
            req = gen.send(res)
    except StopIteration as e:
        return e.value
