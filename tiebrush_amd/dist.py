"""Multi-GPU collapse: input files shard per rank; every rank collapses its own files, group partials are exchanged.

Default protocol — `partials_collapse` (the reference's ancestor is tiewrap.py:96-126, batches of files collapsed
and re-collapsed):
  1. every rank runs the ordinary single-GPU `tbk_collapse_tile` on its own files: local groups with local YC / YX / YD and the
     merge priority (effective end, file, index) of the local representative.  YD is final there: `GSegList::processRead` reads
     only the representative's start and exons (tiebrush.cpp:225-249), which every member of a group shares, the call order
     inside a sample's list is a function of the group keys (:438-457, :511-524), and a sample lives on one rank;
  2. the ranks agree on R-1 coordinate cuts at global tiecov bundle boundaries, found on the local GROUP arrays (all-gather of 64
     sampled group keys, all-reduce rounds as below);
  3. one 48-byte row per local group {tid, pos, strand, n_cigar, effective end, file, index, YC, YX, YD, span, key word} and the local
     representative's CIGAR go to the owner of the group's range (one all-to-all of rows, one of CIGAR words): groups are already
     in coordinate order, so a destination is a contiguous range — nothing is reordered;
  4. the owner merges the R runs it received in output order and reduces equal keys (`tbk_partial_reduce`; or, as the general
     form, `tbk_partial_unpack` + `tbk_collapse_tile` on TieBrush-merged records with explicit priorities): sum YC, sum YX, max YD,
     representative = argmin priority — `SPData::dupAdd` is associative (tiebrush.cpp:408-436) — and runs tiecov on its slice.
Unlike a tiewrap-style hierarchical run, the explicit priority keeps the flat run's representative record.  Exact for integral
YC; inputs that carry a fractional YC (written with --store-frac) fall back, by a collective decision, to the protocol below.

Fallback protocol — `shard_collapse`: the records are shuffled by coordinate, then collapsed once.
Every rank holds some of the input files (a sample lives on one rank).  The ranks agree on R-1 coordinate cuts that no
read of any file spans — global tiecov bundle boundaries — and every record that passes the filters moves to the rank
that owns its coordinate range.  That rank then holds, for its range, the sorted records of ALL input files, i.e. a
tile for the ordinary single-GPU path: one `tbk_collapse_tile` (run-merge sort over the K files as runs, YC / YX / YD
on complete groups) and one `tbk_coverage_tile`, nothing to stitch afterwards.

Why this is exact: groups never span a cut (a group shares (tid,start)), YD chains renew at every cut
(no read of the list reaches across it), tiecov bundles are whole.  The one quantity that depends on records the
filters drop is the effective end of the k-way merge (tmerge.h:28-50: per-file running max, taken BEFORE
passes_options), which decides the representative; the owner of a file computes it and it travels with the record as
its explicit priority (prio_hi = effective end, prio_lo = file << 32 | index in file), which `tbk_collapse_tile`
honours.  `--store-frac` and `-A` are refused here (they need the single-tile path's ordered passes).

Protocol per tile (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" on CPU):
  1. tbk_shard_prepare: per record merge key, filter verdict, effective end, per-file running max of the read ends
  2. all-gather of 64 sampled keys (+ file counts) per rank -> R-1 target cuts
  3. each target moves forward to a global bundle boundary: all-reduce(max) of "farthest read end before the cut",
     all-reduce(min) of "next read start after it", until every cut is clean (tbk_shard_probe_max / _next)
  4. tbk_shard_pack + all-to-all(v): 24-byte rows {tid,pos,strand,n_cigar,effective end,index} and the CIGAR words,
     grouped by (destination, file) so that the receiver sees one sorted run per input file
  5. tbk_shard_unpack -> tbk_collapse_tile over the K runs -> tiecov on the owned slice; junction numbers are offset
     by an all-gather of the per-rank counts.

The rank algorithm is a generator that yields collective requests, so the same code runs over torch.distributed
(`run_distributed`) and over an in-process loopback of R virtual ranks (`run_loopback`, used to exercise R>1 on a single
GPU and in CPU tests).  Arrays are numpy (host tiles, numpy restatement of the four device steps below) or torch CUDA
tensors (tiles resident in HBM: only counts and cut keys ever visit the host).
"""
from __future__ import annotations

import os


import numpy as np

from .soa import SoATile, CovInput

N_SAMPLES = 64
MAX_CUT_ROUNDS = 4096
KEY_INF = 1 << 62        # above every key of a placed read: keys are (tid+1) << 31 | pos+1 (unplaced reads carry exactly this value)


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


class ShardResult:
    """This rank's slice of the global result, in the reference's output order (arrays: numpy or torch).

    n_passed_local: passing input records of THIS rank's files (sum over ranks = inCounter); start / end: 1-based; rep_fidx /
    rep_idx: global file index of the representative record and its index inside that file; cov_input: CovInput (host) or
    DeviceCovView (device), what tiecov reads back for this slice.  After tbk_partial_reduce the representative is known as a ROW of
    the received partials (rep_rows = (rows, row index per group)); tid / rep_fidx / rep_idx are gathered from it on first use."""
    _LAZY = ("tid", "rep_fidx", "rep_idx")

    def __init__(self, n_groups, n_passed_local, tid=None, start=None, end=None, rep_fidx=None, rep_idx=None, yc=None, yx=None, yd=None,
                 cov_input=None, coverage=None, junction_offset=0, n_partials_received=0, rep_rows=None):
        self.n_groups, self.n_passed_local = n_groups, n_passed_local
        self.start, self.end, self.yc, self.yx, self.yd = start, end, yc, yx, yd
        self.cov_input, self.coverage, self.junction_offset, self.n_partials_received = cov_input, coverage, junction_offset, n_partials_received
        self.rep_rows = rep_rows
        if rep_rows is None:
            self.tid, self.rep_fidx, self.rep_idx = tid, rep_fidx, rep_idx

    def __getattr__(self, name):          # (only reached while the lazy fields are not set)
        if name in ShardResult._LAZY and self.__dict__.get("rep_rows") is not None:
            rows, rep = self.rep_rows
            T = _torch()
            rr = rows[rep] if self.n_groups else rows[:0]
            self.tid, self.rep_fidx, self.rep_idx = rr[:, 0], rr[:, 4].to(T.int64), rr[:, 5].to(T.int64) & 0xFFFFFFFF
            return self.__dict__[name]
        raise AttributeError(name)


def _gather_cigars(X, cig_off, cig, rep):
    co = X.u32_to_i64(cig_off)
    ncig = co[rep + 1] - co[rep]
    n = int(rep.shape[0])
    off = X.cat([X.zeros(1, like=ncig), X.cumsum(ncig)]) if n else X.zeros(1, like=co)
    rec_of = X.repeat(X.arange(n, like=co), ncig)
    within = X.arange(int(X.item(off[-1])) if n else 0, like=co) - off[:-1][rec_of]
    idx = co[rep][rec_of] + within
    return ncig, off, cig[idx]


# ---- numpy restatement of the device steps (host tiles: CPU tests, gloo) --------------------------------------------
def _reflen_np(tile):
    ops = np.asarray(tile.cig) & 0xF
    ln = (np.asarray(tile.cig) >> 4).astype(np.int64)
    w = np.where(np.isin(ops, [0, 2, 3, 7, 8]), ln, 0)
    c = np.concatenate([[0], np.cumsum(w)])
    co = np.asarray(tile.cig_off).astype(np.int64)
    return c[co[1:]] - c[co[:-1]]


def _prepare_np(tile, max_nh=2**31 - 1, min_qual=-1, keep_supplementary=False, keep_secondary=False, **_):
    """tbk_shard_prepare on the host: (key, emax, effend, pass)."""
    n = tile.n_records
    flag = np.asarray(tile.flag).astype(np.int64)
    mapped = (flag & 0x4) == 0
    pos = np.asarray(tile.pos).astype(np.int64)
    start = np.where(mapped, pos + 1, 0)
    end = np.where(mapped, pos + _reflen_np(tile), 0)
    tid = np.asarray(tile.tid).astype(np.int64)
    key = np.where(tid < 0, np.int64(1) << 62, ((tid + 1) << 31) | (pos + 1))   # (shard.hip, shard_keys_k)
    nh = np.asarray(tile.nh).astype(np.int64)
    nh = np.where(nh == -(2**31), 0, nh)
    ok = mapped & (np.asarray(tile.mapq).astype(np.int64) >= min_qual) & (nh <= max_nh)
    if not keep_supplementary:
        ok &= (flag & 0x800) == 0
    if not keep_secondary:
        ok &= (flag & 0x100) == 0
    emax = np.zeros(n, np.int64)
    effend = np.zeros(n, np.int64)
    fo = np.asarray(tile.file_off).astype(np.int64)
    for f in range(tile.n_files):
        lo, hi = int(fo[f]), int(fo[f + 1])
        if hi == lo:
            continue
        k = key[lo:hi]
        if np.any(k[1:] < k[:-1]):
            raise ValueError("input not coordinate-sorted")
        emax[lo:hi] = np.maximum.accumulate((k & ~np.int64(0x7FFFFFFF)) | (end[lo:hi] + 1))   # (+ 1: see shard.hip, ShLoad)
        head = np.ones(hi - lo, bool)
        head[1:] = k[1:] != k[:-1]
        run = np.cumsum(head) - 1
        big = np.int64(1) << 33
        effend[lo:hi] = np.maximum.accumulate(run * big + end[lo:hi]) - run * big
    return key, emax, effend.astype(np.int32), ok.astype(np.uint8)


def _probe_max_np(file_off, key, emax, cuts):
    m = np.full(len(cuts), -1, np.int64)
    for f in range(len(file_off) - 1):
        lo, hi = int(file_off[f]), int(file_off[f + 1])
        if hi == lo:
            continue
        i = np.searchsorted(key[lo:hi], cuts, side="left")
        m = np.where(i > 0, np.maximum(m, emax[lo + np.maximum(i - 1, 0)]), m)
    return m


def _probe_next_np(file_off, key, m):
    nxt = np.full(len(m), KEY_INF, np.int64)
    for f in range(len(file_off) - 1):
        lo, hi = int(file_off[f]), int(file_off[f + 1])
        if hi == lo:
            continue
        i = np.searchsorted(key[lo:hi], m, side="right")
        nxt = np.where(i < hi - lo, np.minimum(nxt, key[lo + np.minimum(i, hi - lo - 1)]), nxt)
    return nxt


def _pack_np(tile, key, passm, effend, cuts, world):
    """tbk_shard_pack on the host: (rows, cig words, src_idx, tab[world][k][5])."""
    k = tile.n_files
    fo = np.asarray(tile.file_off).astype(np.int64)
    co = np.asarray(tile.cig_off).astype(np.int64)
    tab = np.zeros((world, k, 5), np.int64)
    order = []
    for d in range(world):
        for f in range(k):
            lo, hi = int(fo[f]), int(fo[f + 1])
            a = lo if d == 0 else lo + int(np.searchsorted(key[lo:hi], cuts[d - 1], side="left"))
            b = hi if d == world - 1 else lo + int(np.searchsorted(key[lo:hi], cuts[d], side="left"))
            idx = np.arange(a, b)[passm[a:b].astype(bool)]
            tab[d, f, 0], tab[d, f, 1] = a, len(idx)
            tab[d, f, 2] = int((co[idx + 1] - co[idx]).sum())
            order.append(idx)
    src = np.concatenate(order) if order else np.zeros(0, np.int64)
    flat = tab.reshape(-1, 5)
    flat[:, 3] = np.concatenate([[0], np.cumsum(flat[:, 1])])[:-1]
    flat[:, 4] = np.concatenate([[0], np.cumsum(flat[:, 2])])[:-1]
    f_of = np.searchsorted(fo, src, side="right") - 1
    ncig = co[src + 1] - co[src]
    rows = np.stack([np.asarray(tile.tid)[src].astype(np.int64), np.asarray(tile.pos)[src].astype(np.int64),
                     np.asarray(tile.strand)[src].astype(np.int64), ncig, effend[src].astype(np.int64), src - fo[f_of]],
                    axis=1).astype(np.int32) if len(src) else np.zeros((0, 6), np.int32)
    _, _, cigw = _gather_cigars(_NP, tile.cig_off, np.asarray(tile.cig), src)
    return rows, cigw, src, tab


def _unpack_np(rows, file_off2):
    n2 = rows.shape[0]
    gf = np.searchsorted(np.asarray(file_off2).astype(np.int64), np.arange(n2), side="right") - 1
    cig_off = np.concatenate([[0], np.cumsum(rows[:, 3].astype(np.int64))]).astype(np.uint32)
    return dict(tid=rows[:, 0].astype(np.int32), pos=rows[:, 1].astype(np.int32), flag=np.zeros(n2, np.uint16),
                mapq=np.full(n2, 255, np.uint8), strand=rows[:, 2].astype(np.uint8), nh=np.full(n2, -(2**31), np.int32),
                cig_off=cig_off, prio_hi=rows[:, 4].astype(np.int64).astype(np.uint64),
                prio_lo=((gf.astype(np.int64) << 32) | rows[:, 5].astype(np.int64)).astype(np.uint64))


def shard_collapse(compute, local_tile: SoATile, first_fidx: int, rank: int, world: int, strategy="cigar",
                   want_coverage=False, device_chain=False, junction_gather=True, **filters):
    """Generator: yields ("all_gather"|"all_reduce_max"|"all_reduce_min"|"exchange_rows"|"all_to_all", payload) requests
    and is sent the result; finally returns a ShardResult.  `compute` provides collapse(tile, **kw) / coverage(cin)
    [/ shard_prepare / shard_probe_* / shard_pack / shard_unpack / groups_to_cov_in / finish_yd] —
    tiebrush_amd.api.Context or a wrapper of it."""
    if filters.get("store_frac") or filters.get("collapse_same"):
        raise ValueError("--store-frac and -A need the single-tile path's ordered passes: single-GPU only (DESIGN.md §7)")
    if strategy in ("full", 1):
        raise ValueError("-L (CIGAR + MD) is single-GPU only: the shuffled rows carry no MD tags")
    X = _xp(local_tile.tid)
    mark = getattr(compute, "mark", None) or (lambda _name: None)       # optional phase hook (tools/prof_dist.py)
    on_dev = _is_t(local_tile.tid) and hasattr(compute, "shard_prepare")
    k = local_tile.n_files
    fo_h = np.asarray(local_tile.file_off).astype(np.int64)
    n = int(fo_h[-1])
    any_tbm = bool(np.asarray(local_tile.tbmerged).any())

    # ---- 1. per-record keys, filter verdict, effective end ---------------------------------------------------
    if on_dev:
        key, emax, effend, passm = compute.shard_prepare(local_tile, **filters)
    else:
        key, emax, effend, passm = _prepare_np(local_tile, **filters)

    mark("prepare")
    # ---- 2. splitter targets from sampled keys (the file table of every rank rides along) --------------------
    idx_h = (np.arange(N_SAMPLES, dtype=np.int64) * n) // N_SAMPLES
    if _is_t(key):                                  # one gather + one small upload; the rest of the payload is host data
        T = _torch()
        meta = T.empty(N_SAMPLES + 3, dtype=T.int64, device=key.device)
        tail = T.tensor([k, first_fidx, int(any_tbm)], dtype=T.int64)
        if n:
            meta[:N_SAMPLES] = key[T.from_numpy(idx_h).to(key.device, non_blocking=True)]
        else:
            meta[:N_SAMPLES] = KEY_INF
        meta[N_SAMPLES:] = tail.to(key.device, non_blocking=True)
    else:
        samp = key[idx_h] if n else np.full(N_SAMPLES, KEY_INF, np.int64)
        meta = np.concatenate([samp, np.array([k, first_fidx, int(any_tbm)], np.int64)])
    allmeta = X.host((yield ("all_gather", meta))).reshape(world, N_SAMPLES + 3)
    ks = allmeta[:, N_SAMPLES].astype(np.int64)
    firsts = allmeta[:, N_SAMPLES + 1].astype(np.int64)
    if not np.array_equal(firsts, np.concatenate([[firsts[0]], firsts[0] + np.cumsum(ks)[:-1]])):
        raise ValueError("ranks must hold consecutive blocks of the input files, in rank order")
    K, kmax = int(ks.sum()), int(ks.max())
    tbm_anywhere = bool(allmeta[:, N_SAMPLES + 2].any())
    p = None
    if world > 1:
        flat = np.sort(allmeta[:, :N_SAMPLES].reshape(-1))
        flat = flat[flat != KEY_INF]
        tgt = np.array([int(flat[(j * len(flat)) // world]) if len(flat) else KEY_INF for j in range(1, world)], np.int64)
        p = _torch().from_numpy(tgt).to(key.device) if _is_t(key) else tgt
        # ---- 3. move every cut forward to a global bundle boundary (all R-1 cuts refined together) ----------
        for _ in range(100000):
            if on_dev:
                m_local = compute.shard_probe_max(fo_h, key, emax, p, X.full(world - 1, -1, like=key))
            else:
                m_local = _probe_max_np(fo_h, X.host(key), X.host(emax), X.host(p))
                m_local = _torch().from_numpy(m_local).to(key.device) if _is_t(key) else m_local
            m = yield ("all_reduce_max", m_local)
            ok = (m < p) | (p == KEY_INF)           # every earlier read of every rank ends before the cut
            if bool(ok.all()):
                break
            if on_dev:
                nxt = compute.shard_probe_next(fo_h, key, m, X.full(world - 1, KEY_INF, like=key))
            else:
                nxt = _probe_next_np(fo_h, X.host(key), X.host(m))
                nxt = _torch().from_numpy(nxt).to(key.device) if _is_t(key) else nxt
            nxt = X.where(ok, p, nxt)
            p = yield ("all_reduce_min", nxt)

    mark("cuts")
    # ---- 4. rows grouped by (destination, file); exchange ----------------------------------------------------
    if on_dev:
        rows, cigw, src, tab = compute.shard_pack(local_tile, key, passm, effend, p, world)
        tab_h = tab.cpu().numpy()
    else:
        hostify = (lambda a: X.host(a)) if _is_t(key) else (lambda a: a)
        rows, cigw, src, tab_h = _pack_np(_host_tile(local_tile), hostify(key), hostify(passm), hostify(effend),
                                          None if p is None else hostify(p), world)
        if _is_t(key):
            T = _torch()
            rows, cigw, src = (T.from_numpy(np.ascontiguousarray(a)).to(key.device) for a in (rows, cigw.view(np.int32), src))
    cnt_rows = tab_h[:, :, 1].sum(1).astype(np.int64)
    cnt_words = tab_h[:, :, 2].sum(1).astype(np.int64)
    n_pass = int(cnt_rows.sum())
    rows, src = rows[:n_pass], src[:n_pass]
    per_file = np.zeros((world, kmax), np.int64)
    per_file[:, :k] = tab_h[:, :, 1]
    mark("pack")
    rrows, rcnt, rcig, rper = yield ("exchange_rows", (rows, cnt_rows, cigw[:int(cnt_words.sum())], cnt_words, per_file))
    ext = None
    if tbm_anywhere:                                # carried YC / YX / YD of TieBrush-merged inputs follow the rows
        if any_tbm and local_tile.yc_in is not None:
            ycb = local_tile.yc_in.view(_torch().int64) if _is_t(local_tile.yc_in) else \
                np.ascontiguousarray(local_tile.yc_in, np.float64).view(np.int64)
            e3 = X.stack([ycb, X.to_i64(local_tile.yx_in), X.to_i64(local_tile.yd_in)], axis=1)[src]
        else:
            e3 = X.stack([X.zeros(n_pass, like=key)] * 3, axis=1) if n_pass else \
                (X.zeros(0, like=key).reshape(0, 3))
        ext, _ = yield ("all_to_all", (e3.contiguous() if _is_t(e3) else np.ascontiguousarray(e3), cnt_rows))
        tflags = np.zeros(kmax, np.int64)
        tflags[:k] = np.asarray(local_tile.tbmerged)
        alltb = X.host((yield ("all_gather", _to_like(tflags, key)))).reshape(world, kmax)

    mark("exchange")
    # ---- 5. the tile of this rank's coordinate range: one sorted run per input file -------------------------
    n2 = int(rrows.shape[0])
    rper_h = np.asarray(rper, np.int64).reshape(world, kmax)
    runs = np.concatenate([rper_h[s, :int(ks[s])] for s in range(world)]) if K else np.zeros(0, np.int64)
    file_off2 = np.zeros(K + 1, np.uint32)
    file_off2[1:] = np.cumsum(runs)
    assert int(file_off2[-1]) == n2
    tbm2 = np.zeros(K, np.uint8)
    if tbm_anywhere:
        tbm2 = np.concatenate([alltb[s, :int(ks[s])] for s in range(world)]).astype(np.uint8)
    if on_dev:
        A = compute.shard_unpack(rrows, file_off2)
    else:
        A = _unpack_np(X.host(rrows) if _is_t(rrows) else np.asarray(rrows), file_off2)
        if _is_t(rrows):
            T = _torch()
            A = {kk: T.from_numpy(np.ascontiguousarray(v.view(np.int64) if v.dtype == np.uint64 else
                                                       v.view(np.int16) if v.dtype == np.uint16 else
                                                       v.view(np.int32) if v.dtype == np.uint32 else v)).to(rrows.device)
                 for kk, v in A.items()}
    tile2 = SoATile(n_files=K, file_off=file_off2, tbmerged=tbm2, tid=A["tid"], pos=A["pos"], flag=A["flag"], mapq=A["mapq"],
                    strand=A["strand"], nh=A["nh"], cig_off=A["cig_off"], cig=rcig, prio_hi=A["prio_hi"], prio_lo=A["prio_lo"])
    if ext is not None:
        tile2.yc_in = ext[:, 0].contiguous().view(_torch().float64) if _is_t(ext) else np.ascontiguousarray(ext[:, 0]).view(np.float64)
        tile2.yx_in = ext[:, 1].contiguous() if _is_t(ext) else np.ascontiguousarray(ext[:, 1])
        tile2.yd_in = ext[:, 2].contiguous() if _is_t(ext) else np.ascontiguousarray(ext[:, 2])
    mark("unpack")
    defer = on_dev and hasattr(compute, "finish_yd")
    fin = compute.collapse(tile2, strategy=strategy, want_coords=True, keep_supplementary=True, keep_secondary=True,
                           **(dict(defer_yd=True) if defer else {}))
    g2 = int(fin["n_groups"])
    rep2 = X.u32_to_i64(fin["rep"]) if _is_t(fin["rep"]) else np.asarray(fin["rep"]).astype(np.int64)
    plo = X.to_i64(A["prio_lo"])[rep2] if _is_t(A["prio_lo"]) else np.asarray(A["prio_lo"]).astype(np.int64)[rep2]
    res = ShardResult(n_groups=g2, n_passed_local=n_pass, tid=tile2.tid[rep2], start=fin["g_start"], end=fin["g_end"],
                      rep_fidx=plo >> 32, rep_idx=plo & 0xFFFFFFFF, yc=fin["yc"], yx=fin["yx"], yd=fin["yd"],
                      n_partials_received=n2)
    mark("collapse")
    # ---- 6. tiecov on the owned slice (whole bundles by construction of the cuts) ---------------------------
    if device_chain:
        res.cov_input = compute.groups_to_cov_in(fin)        # stays in HBM
    else:
        ncg, cof, cg = _gather_cigars(X, tile2.cig_off, tile2.cig, rep2)
        ycf = fin["yc"].to(_torch().float32).to(_torch().float64) if _is_t(fin["yc"]) else \
            np.asarray(fin["yc"]).astype(np.float32).astype(np.float64)
        res.cov_input = CovInput(tid=tile2.tid[rep2], pos=tile2.pos[rep2], flag=X.as_dtype(X.zeros(g2, like=rep2), "u16"),
                                 cig_off=X.as_dtype(cof, "u32"), cig=cg, yc=ycf, strand=tile2.strand[rep2], yx=X.to_i64(fin["yx"]))
    if want_coverage:
        cov = compute.coverage(res.cov_input)
        res.coverage = cov
        if junction_gather:     # (False: the caller carries the counts in its next step's first gather — on EVERY path, so that the ranks
            nj = yield ("all_gather", X.scalar(int(cov["n_junctions"]), like=rep2))   # post the same collectives whichever path a rank took)
            res.junction_offset = int(X.host(nj).reshape(-1)[:rank].sum())
    mark("coverage")
    if defer:
        compute.finish_yd()                                     # the YD column is final now
    mark("finish_yd")
    return res


# ---- group partials (default protocol) ------------------------------------------------------------------------------------------
PROW = 12     # int32 words per partial row (include/tbk.h: TBK_PARTIAL_ROW)


def _partial_keys_np(tile, fin):
    """tbk_partial_keys on the host: (key, emax, not_packable)."""
    ng = int(fin["n_groups"])
    rep = np.asarray(fin["rep"]).astype(np.int64)[:ng]
    tid = np.asarray(tile.tid).astype(np.int64)[rep]
    gs, ge = np.asarray(fin["g_start"]).astype(np.int64)[:ng], np.asarray(fin["g_end"]).astype(np.int64)[:ng]
    key = ((tid + 1) << 31) | gs
    emax = np.maximum.accumulate(((tid + 1) << 31) | (ge + 1)) if ng else np.zeros(0, np.int64)
    yc, yx = np.asarray(fin["yc"])[:ng], np.asarray(fin["yx"]).astype(np.int64)[:ng]
    bad = bool(ng) and bool(np.any(yc != np.rint(yc)) or np.any(yc < 1) or np.any(yc >= 2**31) or np.any(yx < 0) or np.any(yx >= 2**31))
    return key, emax, int(bad)


def _partial_pack_np(tile, fin, key, cuts, world, first_fidx):
    """tbk_partial_pack on the host: (rows [ng, PROW] int32, cig words, tab [world, 3])."""
    ng = int(fin["n_groups"])
    rep = np.asarray(fin["rep"]).astype(np.int64)[:ng]
    fo = np.asarray(tile.file_off).astype(np.int64)
    co = np.asarray(tile.cig_off).astype(np.int64)
    f_of = np.searchsorted(fo, rep, side="right") - 1
    ncig = co[rep + 1] - co[rep]
    rows = np.zeros((ng, PROW), np.int64)
    if ng:
        rows[:, 0] = np.asarray(tile.tid)[rep]
        rows[:, 1] = np.asarray(tile.pos)[rep]
        rows[:, 2] = np.asarray(tile.strand)[rep].astype(np.int64) | (ncig << 8)
        rows[:, 3] = np.asarray(fin["rep_effend"])[:ng]
        rows[:, 4] = first_fidx + f_of
        rows[:, 5] = rep - fo[f_of]
        rows[:, 6] = np.asarray(fin["yc"])[:ng].astype(np.int64)
        rows[:, 7] = np.asarray(fin["yx"])[:ng]
        rows[:, 8] = np.asarray(fin["yd"])[:ng]
        rows[:, 9] = np.asarray(fin["g_end"])[:ng].astype(np.int64) - np.asarray(fin["g_start"])[:ng] + 1
        # (word 10, the key word of tbk_partial_pack, is only read by tbk_partial_reduce: rows packed here are reduced by the general path)
    _, woff, cigw = _gather_cigars(_NP, tile.cig_off, np.asarray(tile.cig), rep)
    b = np.concatenate([[0], np.searchsorted(key, cuts, side="left") if world > 1 else np.zeros(0, np.int64), [ng]]).astype(np.int64)
    tab = np.stack([b[:-1], b[1:] - b[:-1], woff[b[1:]] - woff[b[:-1]]], axis=1).astype(np.int64)
    return rows.astype(np.uint32).view(np.int32) if ng else np.zeros((0, PROW), np.int32), cigw, tab


def _partial_unpack_np(rows):
    r = rows.astype(np.int64)
    n2 = r.shape[0]
    u = r[:, 2] & 0xFFFFFFFF
    cig_off = np.concatenate([[0], np.cumsum(u >> 8)]).astype(np.uint32)
    return dict(tid=r[:, 0].astype(np.int32), pos=r[:, 1].astype(np.int32), flag=np.zeros(n2, np.uint16), mapq=np.full(n2, 255, np.uint8),
                strand=(u & 0xFF).astype(np.uint8), nh=np.full(n2, -(2**31), np.int32), cig_off=cig_off,
                yc_in=(r[:, 6] & 0xFFFFFFFF).astype(np.float64), yx_in=r[:, 7].copy(), yd_in=r[:, 8].copy(),
                prio_hi=r[:, 3].astype(np.uint64), prio_lo=((r[:, 4] << 32) | (r[:, 5] & 0xFFFFFFFF)).astype(np.uint64))


def _partials_rounds(compute, X, mark, on_dev, local_tile, fin, first_fidx, rank, world, strategy, stats, filters):
    """steps 2 - 4 of the protocol in its first form: cuts walked forward in all-reduce rounds, counts exchanged before the rows.
    Returns (rrows, rcnt, rcig) or "shuffle"."""
    k = local_tile.n_files
    ng = int(fin["n_groups"])
    # ---- 2. cut keys of the local groups; one all-gather: samples + file table + "can this rank's partials be packed" -----
    if on_dev:
        key, emax, bad = compute.partial_keys(local_tile, fin)
    else:
        key, emax, bad = _partial_keys_np(_host_tile(local_tile), {kk: (X.host(v) if _is_t(v) else v) for kk, v in fin.items()
                                                                   if kk in ("n_groups", "rep", "yc", "yx", "g_start", "g_end")})
        if _is_t(local_tile.tid):
            key, emax = (_torch().from_numpy(a).to(local_tile.tid.device) for a in (key, emax))
    idx_h = (np.arange(N_SAMPLES, dtype=np.int64) * ng) // N_SAMPLES
    tail = np.array([k, first_fidx, bad], np.int64)
    if _is_t(key):
        T = _torch()
        meta = T.empty(N_SAMPLES + 3, dtype=T.int64, device=key.device)
        if ng:
            meta[:N_SAMPLES] = key[T.from_numpy(idx_h).to(key.device, non_blocking=True)]
        else:
            meta[:N_SAMPLES] = KEY_INF
        meta[N_SAMPLES:] = T.from_numpy(tail).to(key.device, non_blocking=True)
    else:
        meta = np.concatenate([key[idx_h] if ng else np.full(N_SAMPLES, KEY_INF, np.int64), tail])
    allmeta = X.host((yield ("all_gather", meta))).reshape(world, N_SAMPLES + 3)
    ks = allmeta[:, N_SAMPLES].astype(np.int64)
    firsts = allmeta[:, N_SAMPLES + 1].astype(np.int64)
    if not np.array_equal(firsts, np.concatenate([[firsts[0]], firsts[0] + np.cumsum(ks)[:-1]])):
        raise ValueError("ranks must hold consecutive blocks of the input files, in rank order")
    if bool(allmeta[:, N_SAMPLES + 2].any()) and strategy in ("full", 1):
        raise ValueError("-L with carried fractional YC: the record shuffle carries no MD strings (run on one GPU)")
    if bool(allmeta[:, N_SAMPLES + 2].any()):
        # a carried fractional YC somewhere (or a count beyond 31 bits): sums across ranks would not keep the reference's order of
        # additions — every rank takes the record shuffle for this tile (the decision is collective: same data on all ranks)
        return "shuffle"
    fo1 = np.array([0, ng], np.int64)
    p = None
    if world > 1:
        flat = np.sort(allmeta[:, :N_SAMPLES].reshape(-1))
        flat = flat[flat != KEY_INF]
        tgt = np.array([int(flat[(j * len(flat)) // world]) if len(flat) else KEY_INF for j in range(1, world)], np.int64)
        p = _torch().from_numpy(tgt).to(key.device) if _is_t(key) else tgt
        # ---- 3. every cut moves forward to a global bundle boundary of the GROUPS (= of the passing records) ------------
        # (a round moves every unsettled cut to the next read start beyond what reaches across it: a cut settles as soon as it
        # meets a gap, so the rounds are bounded by the bundles a cut has to cross — a handful on real data; MAX_CUT_ROUNDS is a
        # guard against an input that is one bundle from end to end, which cannot be cut at all)
        for rnd in range(MAX_CUT_ROUNDS + 1):
            if rnd == MAX_CUT_ROUNDS:
                raise ValueError("no bundle boundary within %d rounds of a coordinate cut: the input cannot be sharded by range" % MAX_CUT_ROUNDS)
            if stats is not None:
                stats["cut_rounds"] = rnd + 1
            if on_dev:
                m_local = compute.shard_probe_max(fo1, key, emax, p, X.full(world - 1, -1, like=key))
            else:
                m_local = _probe_max_np(fo1, X.host(key), X.host(emax), X.host(p))
                m_local = _torch().from_numpy(m_local).to(key.device) if _is_t(key) else m_local
            m = yield ("all_reduce_max", m_local)
            ok = (m < p) | (p == KEY_INF)
            if bool(ok.all()):
                break
            if on_dev:
                nxt = compute.shard_probe_next(fo1, key, m, X.full(world - 1, KEY_INF, like=key))
            else:
                nxt = _probe_next_np(fo1, X.host(key), X.host(m))
                nxt = _torch().from_numpy(nxt).to(key.device) if _is_t(key) else nxt
            nxt = X.where(ok, p, nxt)
            p = yield ("all_reduce_min", nxt)
    mark("cuts")
    # ---- 4. rows + CIGAR words in group order; exchange ----------------------------------------------------------------------
    if on_dev:
        rows, cigw, tab = compute.partial_pack(local_tile, fin, key, p, world, first_fidx, strategy=strategy, **filters)
        tab_h = tab.cpu().numpy()
    else:
        hostify = (lambda a: X.host(a)) if _is_t(key) else (lambda a: a)
        hfin = {kk: (X.host(v) if _is_t(v) else v) for kk, v in fin.items() if not kk.startswith("_")}
        rows, cigw, tab_h = _partial_pack_np(_host_tile(local_tile), hfin, hostify(key), None if p is None else hostify(p), world, first_fidx)
        if _is_t(key):
            T = _torch()
            rows, cigw = (T.from_numpy(np.ascontiguousarray(a)).to(key.device) for a in (rows, cigw.view(np.int32)))
    cnt_rows = tab_h[:, 1].astype(np.int64)
    cnt_words = tab_h[:, 2].astype(np.int64)
    if stats is not None:
        stats["wire_rows"] = int(cnt_rows.sum())
        stats["wire_bytes"] = int(cnt_rows.sum()) * PROW * 4 + int(cnt_words.sum()) * 4
        stats["wire_bytes_off_rank"] = stats["wire_bytes"] - (int(cnt_rows[rank]) * PROW * 4 + int(cnt_words[rank]) * 4)
    mark("pack")
    mdb = None
    if strategy in ("full", 1):                         # -L: the MD strings of the rows, in row order (the rows' word 11 says how long)
        mdb, mdtab = compute.partial_pack_md(local_tile, fin, tab, world, rows)
        mdc = mdtab.cpu().numpy().astype(np.int64)
    rrows, rcnt, rcig, _ = yield ("exchange_rows", (rows[:ng], cnt_rows, cigw[:int(cnt_words.sum())], cnt_words, cnt_rows.reshape(world, 1)))
    rmd = None
    if mdb is not None:
        rmd, _ = yield ("all_to_all", (mdb[:int(mdc.sum())], mdc))
    mark("exchange")
    return rrows, rcnt, rcig, rmd


def _owner_reduce_fast(compute, X, rrows, file_off2, rcig, device_chain, strategy, rmd=None):
    """tbk_partial_reduce on the rows as they arrived; None when it hands the tile to the general path (a hashed key word shared by two
    alignments, a pile-up of partials, more runs than a window takes)"""
    if not (hasattr(compute, "partial_reduce") and os.environ.get("TBK_PARTIAL_REDUCE", "1") != "0"):
        return None
    from ._lib import TbkError
    try:
        return compute.partial_reduce(rrows, file_off2, rcig, want_view=device_chain, strategy=strategy, **(dict(md=rmd) if rmd is not None else {}))
    except TbkError as e:
        if e.status not in (-8, -4, -5):
            raise
        return None


def _partials_lists(compute, local_tile, fin, first_fidx, rank, world, strategy, want_coverage, device_chain, stats, carry, junction_gather, filters):
    """The group-partials protocol with the cut search on lists (tbk_partial_stage_*, include/tbk.h): three all-gathers of small device
    arrays, ONE read-back (the gathered exchange table with every verdict in its flag words), two all-to-alls whose counts every rank
    already knows.  Returns a ShardResult, or the name of the protocol that has to take the tile instead ("rounds": a cut no list could
    settle; "shuffle": partials that cannot be packed) — a collective decision, every rank reads the same table."""
    X = _xp(local_tile.tid)
    mark = getattr(compute, "mark", None) or (lambda _name: None)
    ng = int(fin["n_groups"])
    n_pass = int(fin["n_passed"])
    key, emax, meta = compute.partial_stage_keys(local_tile, fin, first_fidx, carry)
    allmeta = yield ("all_gather", meta)
    targets = allc = None
    if world > 1:
        targets, cands = compute.partial_stage_cands(key, emax, allmeta, world)
        allc = yield ("all_gather", cands)
    rows, cigw, tabx, _cuts = compute.partial_stage_pack(local_tile, fin, key, meta, allc, targets, world, first_fidx, strategy=strategy, **filters)
    full = strategy in ("full", 1)
    mdb = None
    if full:            # -L: the representatives' MD strings ride beside the rows; their byte counts join the gathered table
        mdb, mdtab = compute.partial_pack_md(local_tile, fin, tabx, world, rows)
        tabx = _torch().cat([tabx, mdtab])
    ntx = world * 3 + 4 + (world if full else 0)
    alltab = np.asarray(X.host((yield ("all_gather", tabx)))).reshape(world, ntx).astype(np.int64)     # the one read-back
    mdcnt = alltab[:, world * 3 + 4:] if full else None
    alltab = alltab[:, :world * 3 + 4]
    if stats is not None:
        stats["collectives"] = stats.get("collectives", 0) + (3 if world > 1 else 2)
        stats["host_syncs"] = stats.get("host_syncs", 0) + 1
        stats["cut_rounds"] = 0
        stats["prev_carry"] = alltab[:, world * 3 + 3].copy()
    flags, ks, firsts = alltab[:, world * 3], alltab[:, world * 3 + 1], alltab[:, world * 3 + 2]
    if not np.array_equal(firsts, np.concatenate([[firsts[0]], firsts[0] + np.cumsum(ks)[:-1]])):
        raise ValueError("ranks must hold consecutive blocks of the input files, in rank order")
    if (flags >> 8).any():
        raise RuntimeError("device error bits 0x%x in the partials' pack" % int((flags >> 8).max()))
    if (flags & 1).any():
        return "shuffle"
    if (flags & 2).any():
        return "rounds"
    mark("pack")
    tab = alltab[:, :world * 3].reshape(world, world, 3)
    send_rows, send_words = tab[rank, :, 1].copy(), tab[rank, :, 2].copy()
    recv_rows, recv_words = tab[:, rank, 1].copy(), tab[:, rank, 2].copy()
    if stats is not None:
        stats["wire_rows"] = int(send_rows.sum())
        stats["wire_bytes"] = int(send_rows.sum()) * PROW * 4 + int(send_words.sum()) * 4
        stats["wire_bytes_off_rank"] = stats["wire_bytes"] - (int(send_rows[rank]) * PROW * 4 + int(send_words[rank]) * 4)
        stats["collectives"] += 2
    big = (int(tab[:, :, 1].max()), int(tab[:, :, 2].max()))
    rrows, rcig = yield ("exchange_known", (rows[:ng], send_rows, recv_rows, cigw[:int(send_words.sum())], send_words, recv_words, big))
    rmd = None
    if full:
        smd, rmdc = mdcnt[rank, :].copy(), mdcnt[:, rank].copy()
        rmd, _ = yield ("exchange_known", (mdb[:int(smd.sum())], smd, rmdc, mdb[:0], np.zeros(world, np.int64), np.zeros(world, np.int64),
                                           (int(mdcnt.max()), 0)))
        if stats is not None:
            stats["collectives"] += 1
            stats["wire_bytes"] += int(smd.sum())
    mark("exchange")
    n2 = int(rrows.shape[0])
    file_off2 = np.zeros(world + 1, np.uint32)
    file_off2[1:] = np.cumsum(recv_rows)
    assert int(file_off2[-1]) == n2
    fast = _owner_reduce_fast(compute, X, rrows, file_off2, rcig, device_chain, strategy, rmd)
    if stats is not None:
        stats["host_syncs"] += 1            # (the owner's reduce reads its group count back)
    if fast is None:
        return ("general", rrows, recv_rows, rcig, rmd)
    T = _torch()
    g2 = int(fast["n_groups"])
    res = ShardResult(n_groups=g2, n_passed_local=n_pass, start=fast["g_start"], end=fast["g_end"], yc=fast["yc"], yx=fast["yx"], yd=fast["yd"],
                      n_partials_received=n2, rep_rows=(rrows, fast["rep"]))
    mark("reduce")
    rep2 = fast["rep"]
    if device_chain:
        res.cov_input = fast["view"]
    else:
        rep2 = fast["rep"].to(T.int64)
        rr = rrows[rep2] if g2 else rrows[:0]
        A = compute.partial_unpack(rrows)
        ncg, cof, cg = _gather_cigars(X, A["cig_off"], rcig, rep2)
        res.cov_input = CovInput(tid=rr[:, 0].contiguous(), pos=rr[:, 1].contiguous(), flag=X.as_dtype(X.zeros(g2, like=rep2), "u16"),
                                 cig_off=X.as_dtype(cof, "u32"), cig=cg, yc=fast["yc"].to(T.float32).to(T.float64),
                                 strand=A["strand"][rep2], yx=fast["yx"])
    if want_coverage:
        cov = compute.coverage(res.cov_input)
        res.coverage = cov
        if junction_gather:
            nj = yield ("all_gather", X.scalar(int(cov["n_junctions"]), like=rep2))
            res.junction_offset = int(X.host(nj).reshape(-1)[:rank].sum())
            if stats is not None:
                stats["collectives"] += 1
    mark("coverage")
    return res


def partials_collapse(compute, local_tile: SoATile, first_fidx: int, rank: int, world: int, strategy="cigar", want_coverage=False,
                      device_chain=False, local=None, stats=None, cut_search="lists", carry=0, junction_gather=True, **filters):
    """Generator like shard_collapse (same requests, same ShardResult): collapse locally, exchange group partials, reduce by key
    on the owner.  `local`: the result of the local collapse when the driver has already run it (bench.py collapses tile i + 1
    while tile i is exchanged) — a dict of compute.collapse(..., want_coords=True, want_effend=True) with a final `yd`.
    `stats` (dict, optional) receives wire_rows / wire_bytes of this rank's exchange."""
    if filters.get("store_frac") or filters.get("collapse_same"):
        raise ValueError("--store-frac and -A need the single-tile path's ordered passes: single-GPU only (DESIGN.md §7)")
    X = _xp(local_tile.tid)
    mark = getattr(compute, "mark", None) or (lambda _name: None)
    on_dev = _is_t(local_tile.tid) and hasattr(compute, "partial_keys")
    full = strategy in ("full", 1)
    if full and not (on_dev and hasattr(compute, "partial_pack_md")):
        raise ValueError("-L (CIGAR + MD) across ranks needs device-resident tiles: the MD strings travel through tbk_partial_pack_md")
    k = local_tile.n_files
    # ---- 1. the ordinary single-GPU collapse of this rank's files ---------------------------------------------------------
    fin = local
    if fin is None:
        fin = compute.collapse(local_tile, strategy=strategy, want_coords=True, want_effend=True, **filters)
    ng = int(fin["n_groups"])
    n_pass = int(fin["n_passed"])
    mark("local")
    general = None
    if on_dev and hasattr(compute, "partial_stage_keys") and cut_search == "lists":
        got = yield from _partials_lists(compute, local_tile, fin, first_fidx, rank, world, strategy, want_coverage, device_chain, stats, carry,
                                         junction_gather, filters)
        if isinstance(got, ShardResult):
            return got
        if got == "shuffle":
            res = yield from shard_collapse(compute, local_tile, first_fidx, rank, world, strategy=strategy, want_coverage=want_coverage,
                                            device_chain=device_chain, junction_gather=junction_gather, **filters)
            return res
        if isinstance(got, tuple):          # the rows are here, the owner's merge-reduce handed them to the general path
            general = got
        # ("rounds": a cut no list settled — the walk below takes the tile from the start)
    if general is None:
        got = yield from _partials_rounds(compute, X, mark, on_dev, local_tile, fin, first_fidx, rank, world, strategy, stats, filters)
        if got == "shuffle":
            res = yield from shard_collapse(compute, local_tile, first_fidx, rank, world, strategy=strategy, want_coverage=want_coverage,
                                            device_chain=device_chain, junction_gather=junction_gather, **filters)
            return res
        rrows, rcnt, rcig, rmd = got
    else:
        rrows, rcnt, rcig, rmd = general[1], general[2], general[3], general[4]
    # ---- 5. the partials of this rank's coordinate range: one run per source rank, TieBrush-merged, explicit priorities ----
    n2 = int(rrows.shape[0])
    file_off2 = np.zeros(world + 1, np.uint32)
    file_off2[1:] = np.cumsum(np.asarray(rcnt, np.int64))
    assert int(file_off2[-1]) == n2
    fast = None
    if on_dev and general is None:                      # the owner's merge-reduce on the rows as they arrived
        fast = _owner_reduce_fast(compute, X, rrows, file_off2, rcig, device_chain, strategy, rmd)
    if fast is not None:
        T = _torch()
        g2 = int(fast["n_groups"])
        res = ShardResult(n_groups=g2, n_passed_local=n_pass, start=fast["g_start"], end=fast["g_end"], yc=fast["yc"], yx=fast["yx"], yd=fast["yd"],
                          n_partials_received=n2, rep_rows=(rrows, fast["rep"]))
        mark("reduce")
        rep2 = fast["rep"]
        if device_chain:
            res.cov_input = fast["view"]
        else:
            rep2 = fast["rep"].to(T.int64)
            rr = rrows[rep2] if g2 else rrows[:0]
            A = compute.partial_unpack(rrows)
            ncg, cof, cg = _gather_cigars(X, A["cig_off"], rcig, rep2)
            res.cov_input = CovInput(tid=rr[:, 0].contiguous(), pos=rr[:, 1].contiguous(), flag=X.as_dtype(X.zeros(g2, like=rep2), "u16"),
                                     cig_off=X.as_dtype(cof, "u32"), cig=cg, yc=fast["yc"].to(T.float32).to(T.float64),
                                     strand=A["strand"][rep2], yx=fast["yx"])
        if want_coverage:
            cov = compute.coverage(res.cov_input)
            res.coverage = cov
            if junction_gather:
                nj = yield ("all_gather", X.scalar(int(cov["n_junctions"]), like=rep2))
                res.junction_offset = int(X.host(nj).reshape(-1)[:rank].sum())
        mark("coverage")
        return res
    if on_dev:
        A = compute.partial_unpack(rrows)
    else:
        A = _partial_unpack_np(X.host(rrows) if _is_t(rrows) else np.asarray(rrows))
        if _is_t(rrows):
            T = _torch()
            sg = {np.dtype(np.uint64): np.int64, np.dtype(np.uint32): np.int32, np.dtype(np.uint16): np.int16}
            A = {kk: T.from_numpy(np.ascontiguousarray(v.view(sg.get(v.dtype, v.dtype)))).to(rrows.device) for kk, v in A.items()}
    tile2 = SoATile(n_files=world, file_off=file_off2, tbmerged=np.ones(world, np.uint8), tid=A["tid"], pos=A["pos"], flag=A["flag"],
                    mapq=A["mapq"], strand=A["strand"], nh=A["nh"], cig_off=A["cig_off"], cig=rcig, yc_in=A["yc_in"], yx_in=A["yx_in"],
                    yd_in=A["yd_in"], prio_hi=A["prio_hi"], prio_lo=A["prio_lo"])
    if full:                                            # -L: the partials' MD strings are the tile's MD columns
        tile2.md_off, tile2.md_has = compute.partial_unpack_md(rrows)
        tile2.md = rmd
    mark("unpack")
    fin2 = compute.collapse(tile2, strategy=strategy, want_coords=True, keep_supplementary=True, keep_secondary=True)
    g2 = int(fin2["n_groups"])
    rep2 = X.u32_to_i64(fin2["rep"]) if _is_t(fin2["rep"]) else np.asarray(fin2["rep"]).astype(np.int64)
    plo = X.to_i64(A["prio_lo"])[rep2] if _is_t(A["prio_lo"]) else np.asarray(A["prio_lo"]).astype(np.int64)[rep2]
    res = ShardResult(n_groups=g2, n_passed_local=n_pass, tid=tile2.tid[rep2], start=fin2["g_start"], end=fin2["g_end"],
                      rep_fidx=plo >> 32, rep_idx=plo & 0xFFFFFFFF, yc=fin2["yc"], yx=fin2["yx"], yd=fin2["yd"], n_partials_received=n2)
    mark("reduce")
    # ---- 6. tiecov on the owned slice (whole bundles by construction of the cuts) -------------------------------------------
    if device_chain:
        res.cov_input = compute.groups_to_cov_in(fin2)
    else:
        ncg, cof, cg = _gather_cigars(X, tile2.cig_off, tile2.cig, rep2)
        ycf = fin2["yc"].to(_torch().float32).to(_torch().float64) if _is_t(fin2["yc"]) else \
            np.asarray(fin2["yc"]).astype(np.float32).astype(np.float64)
        res.cov_input = CovInput(tid=tile2.tid[rep2], pos=tile2.pos[rep2], flag=X.as_dtype(X.zeros(g2, like=rep2), "u16"),
                                 cig_off=X.as_dtype(cof, "u32"), cig=cg, yc=ycf, strand=tile2.strand[rep2], yx=X.to_i64(fin2["yx"]))
    if want_coverage:
        cov = compute.coverage(res.cov_input)
        res.coverage = cov
        if junction_gather:     # (False: the caller carries the counts in its next step's first gather — on EVERY path, so that the ranks
            nj = yield ("all_gather", X.scalar(int(cov["n_junctions"]), like=rep2))   # post the same collectives whichever path a rank took)
            res.junction_offset = int(X.host(nj).reshape(-1)[:rank].sum())
    mark("coverage")
    return res


def _to_like(a, like):
    return _torch().from_numpy(np.ascontiguousarray(a)).to(like.device) if _is_t(like) else a


def _host_tile(tile):
    """numpy view of a tile whose arrays may be torch tensors (host restatement path only)."""
    if not _is_t(tile.tid):
        return tile
    h = lambda a: None if a is None else a.cpu().numpy()
    return SoATile(n_files=tile.n_files, file_off=tile.file_off, tbmerged=tile.tbmerged, tid=h(tile.tid), pos=h(tile.pos),
                   flag=h(tile.flag).view(np.uint16), mapq=h(tile.mapq), strand=h(tile.strand), nh=h(tile.nh),
                   cig_off=h(tile.cig_off).view(np.uint32), cig=h(tile.cig).view(np.uint32))


# ---- drivers ---------------------------------------------------------------------------------------------------
def run_loopback(compute, tiles, first_fidx, per_rank=None, **kw):
    """Run R virtual ranks in one process: steps the R generators in lockstep and serves their collectives.
    mode="partials" (default): collapse locally, exchange group partials; mode="shuffle": the record shuffle.
    per_rank: optional list of R dicts of keyword arguments that differ by rank (local=..., stats=...)."""
    world = len(tiles)
    gen_fn = shard_collapse if kw.pop("mode", "partials") == "shuffle" else partials_collapse
    gens = [gen_fn(compute, tiles[r], first_fidx[r], r, world, **kw, **(per_rank[r] if per_rank else {})) for r in range(world)]
    reqs = [next(g) for g in gens]
    results = [None] * world
    while any(r is None for r in results):
        assert all(res is None for res in results) and len({q[0] for q in reqs}) == 1, "ranks diverged"
        kind = reqs[0][0]
        pay = [q[1] for q in reqs]
        X = _xp(pay[0][0] if kind in ("exchange_rows", "all_to_all", "exchange_known") else pay[0])
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
                    ch = np.asarray(X.host(cnt) if _is_t(cnt) else cnt, np.int64)
                    o = int(ch[:d].sum())
                    parts.append(data[o:o + int(ch[d])])
                    cnts.append(int(ch[d]))
                out.append((X.cat(parts), np.array(cnts, np.int64)))
        elif kind == "exchange_known":       # every rank knows every count already: rows and CIGAR words move, nothing else
            out = []
            for d in range(world):
                rows, words = [], []
                for s_ in range(world):
                    R, sr, _rr, cig, sw, _rw, _big = pay[s_]
                    o, oc = int(np.asarray(sr)[:d].sum()), int(np.asarray(sw)[:d].sum())
                    rows.append(R[o:o + int(sr[d])])
                    words.append(cig[oc:oc + int(sw[d])])
                assert [int(r.shape[0]) for r in rows] == [int(c) for c in pay[d][2]]
                out.append((X.cat(rows), X.cat(words)))
        elif kind == "exchange_rows":
            out = []
            for d in range(world):
                rows, words, cnts, metas = [], [], [], []
                for s in range(world):
                    R, cnt, cig, ccnt, per = pay[s]
                    ch, cc = np.asarray(cnt, np.int64), np.asarray(ccnt, np.int64)
                    o, oc = int(ch[:d].sum()), int(cc[:d].sum())
                    rows.append(R[o:o + int(ch[d])])
                    words.append(cig[oc:oc + int(cc[d])])
                    cnts.append(int(ch[d]))
                    metas.append(np.asarray(per)[d])
                out.append((X.cat(rows), np.array(cnts, np.int64), X.cat(words), np.stack(metas)))
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
    if dist.get_backend(group) == "gloo":
        dev = "cpu"

    _signed = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64, np.dtype(np.uint16): np.int16}

    stage = dist.get_backend(group) == "gloo"       # gloo moves host memory: device tensors are staged through the host

    def t(a):
        if _is_t(a):
            return a.cpu() if (stage and a.is_cuda) else a
        a = np.ascontiguousarray(a)
        if a.dtype in _signed:          # collectives have no unsigned types: ship the same bits as signed
            a = a.view(_signed[a.dtype])
        return torch.from_numpy(a).to(dev)

    def back(x, like):
        if _is_t(like):
            return x.to(like.device) if x.device != like.device else x
        r = x.cpu().numpy()
        return r.view(like.dtype) if np.asarray(like).dtype in _signed else r

    A2A_MAX_BYTES = int(os.environ.get("TBK_A2A_MAX_BYTES", 256 << 20))   # per-peer piece of one all_to_all_single call

    def a2a_rows(x, send_cnt, recv_cnt, big=None):
        """all_to_all of the row blocks of x (dim 0 split by send_cnt), recv_cnt rows from each source.  One rank: a
        copy.  Blocks beyond A2A_MAX_BYTES go out in several rounds: the RCCL of this image drops the tail of a
        1.5 GB self-exchange (a probe of round 4: 64 M rows x 24 B arrive as 32 M rows + zeros), and rounds of a
        bounded size also bound the staging memory."""
        out = torch.empty((int(sum(recv_cnt)),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if world == 1:
            out.copy_(x[:out.shape[0]])
            return out
        row_bytes = max(1, x.element_size() * int(np.prod(x.shape[1:], dtype=np.int64)))
        chunk = max(1, A2A_MAX_BYTES // row_bytes)
        if big is None:        # the largest piece any rank sends: agreed on by a reduction — or known to the caller (`big`)
            big = max(int(max(send_cnt)), int(max(recv_cnt)))
            rounds = torch.tensor([(big + chunk - 1) // chunk], dtype=torch.int64, device=x.device if not stage else "cpu")
            dist.all_reduce(rounds, op=dist.ReduceOp.MAX, group=group)
            rounds = max(1, int(rounds))
        else:
            if int(big) == 0:            # nothing moves anywhere (every rank knows): no collective
                return out
            rounds = max(1, (int(big) + chunk - 1) // chunk)
        if rounds == 1:
            dist.all_to_all_single(out, x, output_split_sizes=[int(c) for c in recv_cnt], input_split_sizes=[int(c) for c in send_cnt],
                                   group=group)
            return out
        so = np.concatenate([[0], np.cumsum(send_cnt)]).astype(np.int64)
        ro = np.concatenate([[0], np.cumsum(recv_cnt)]).astype(np.int64)
        for r in range(rounds):
            ss = [max(0, min(chunk, int(send_cnt[d]) - r * chunk)) for d in range(world)]
            rs = [max(0, min(chunk, int(recv_cnt[s_]) - r * chunk)) for s_ in range(world)]
            xin = torch.cat([x[int(so[d]) + r * chunk: int(so[d]) + r * chunk + ss[d]] for d in range(world)])
            tmp = torch.empty((sum(rs),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_to_all_single(tmp, xin, output_split_sizes=rs, input_split_sizes=ss, group=group)
            o = 0
            for s_ in range(world):
                out[int(ro[s_]) + r * chunk: int(ro[s_]) + r * chunk + rs[s_]] = tmp[o:o + rs[s_]]
                o += rs[s_]
        return out

    gen_fn = shard_collapse if kw.pop("mode", "partials") == "shuffle" else partials_collapse
    gen = gen_fn(compute, tile, first_fidx, rank, world, **kw)
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
                sc_h = (c.cpu().numpy() if _is_t(cnt) else np.asarray(cnt)).astype(np.int64)
                rc_h = rc.cpu().numpy().astype(np.int64)
                out = a2a_rows(t(data).contiguous(), sc_h.tolist(), rc_h.tolist())
                res = (back(out, data), rc_h)
            elif kind == "exchange_known":
                R, sr, rr, cig, sw, rw, big = pay
                outR = a2a_rows(t(R).contiguous(), [int(c) for c in sr], [int(c) for c in rr], big=big[0])
                outC = a2a_rows(t(cig).contiguous(), [int(c) for c in sw], [int(c) for c in rw], big=big[1])
                res = (back(outR, R), back(outC, cig))
            elif kind == "exchange_rows":
                R, cnt, cig, ccnt, per = pay
                # one small all-to-all tells every rank what it will receive: [rows, CIGAR words, rows per file of the sender]
                sc_h = np.concatenate([np.asarray(cnt, np.int64).reshape(world, 1), np.asarray(ccnt, np.int64).reshape(world, 1),
                                       np.asarray(per, np.int64).reshape(world, -1)], axis=1)
                c = t(sc_h).contiguous()
                rc = torch.empty_like(c)
                dist.all_to_all_single(rc, c, group=group)
                rc_h = rc.cpu().numpy().astype(np.int64)
                outR = a2a_rows(t(R).contiguous(), sc_h[:, 0].tolist(), rc_h[:, 0].tolist())
                outC = a2a_rows(t(cig).contiguous(), sc_h[:, 1].tolist(), rc_h[:, 1].tolist())
                res = (back(outR, R), rc_h[:, 0].copy(), back(outC, cig), rc_h[:, 2:].copy())
            else:
                raise AssertionError(kind)
            req = gen.send(res)
    except StopIteration as e:
        return e.value
