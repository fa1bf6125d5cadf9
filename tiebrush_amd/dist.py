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
     few all-reduce(max/min) rounds of two scalars — so every rank ends up owning whole tiecov bundles
  4. all-to-all(v) of the partial groups {tid,pos,strand,yx,yd,prio,ncig | yc | CIGAR words}
  5. stitch = tbk_collapse_tile again over the received partials ("files" = source ranks, all marked
     TieBrush-merged so YC/YX/YD are carried; representative by explicit priority)
  6. tiecov on the owned slice; junction numbers are offset by an all-gather of the per-rank counts.
The rank algorithm is a generator that yields collective requests, so the same code runs over
torch.distributed (`run_distributed`) and over an in-process loopback of R virtual ranks (`run_loopback`,
used to exercise R>1 on a single GPU and in CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .soa import SoATile, CovInput

N_SAMPLES = 64
KEY_INF = np.iinfo(np.int64).max


@dataclass
class ShardResult:
    """This rank's slice of the global result, in the reference's output order."""
    n_groups: int
    n_passed_local: int          # passing input records of THIS rank's files (sum over ranks = inCounter)
    tid: np.ndarray
    start: np.ndarray            # 1-based
    end: np.ndarray
    rep_fidx: np.ndarray         # global file index of the representative record
    rep_idx: np.ndarray          # its index inside that file
    yc: np.ndarray
    yx: np.ndarray
    yd: np.ndarray
    cov_input: CovInput          # what tiecov reads back for this slice
    coverage: Optional[dict] = None
    junction_offset: int = 0


def _key64(tid, pos1):
    return ((np.asarray(tid, np.int64) + 1) << 32) | np.asarray(pos1, np.int64)


def _gather_cigars(tile: SoATile, rep):
    rep = np.asarray(rep, np.int64)
    co = np.asarray(tile.cig_off, np.int64)
    ncig = co[rep + 1] - co[rep]
    off = np.zeros(len(rep) + 1, np.int64)
    np.cumsum(ncig, out=off[1:])
    tot = int(off[-1])
    rec_of = np.repeat(np.arange(len(rep)), ncig)
    within = np.arange(tot) - off[:-1][rec_of]
    cig = np.asarray(tile.cig)[co[rep][rec_of] + within] if tot else np.zeros(0, np.uint32)
    return ncig, off, cig.astype(np.uint32)


def shard_collapse(compute, local_tile: SoATile, first_fidx: int, rank: int, world: int, strategy="cigar",
                   want_coverage=False, **filters):
    """Generator: yields ("all_gather"|"all_reduce_max"|"all_reduce_min"|"all_to_all", payload) requests and is
    sent the result; finally returns a ShardResult.  `compute` provides collapse(tile, **kw) / coverage(cin)
    returning host (numpy) dicts — tiebrush_amd.api.Context on a GPU."""
    if filters.get("store_frac") or filters.get("collapse_same"):
        raise ValueError("--store-frac and -A need a global second pass: single-GPU only (DESIGN.md §7)")
    loc = compute.collapse(local_tile, strategy=strategy, want_coords=True, want_effend=True, **filters)
    ng = int(loc["n_groups"])
    rep = np.asarray(loc["rep"]).astype(np.int64)
    fo = np.asarray(local_tile.file_off, np.int64)
    lf = np.searchsorted(fo, rep, side="right") - 1 if ng else np.zeros(0, np.int64)
    tid = np.asarray(local_tile.tid)[rep].astype(np.int64) if ng else np.zeros(0, np.int64)
    gstart = np.asarray(loc["g_start"]).astype(np.int64)
    gend = np.asarray(loc["g_end"]).astype(np.int64)
    key = _key64(tid, gstart)                       # nondecreasing: local output order is bucket order
    ekey = _key64(tid, gend)
    emax = np.maximum.accumulate(ekey) if ng else ekey

    # ---- 2. splitter targets from sampled keys -------------------------------------------------------------
    samp = np.full(N_SAMPLES, KEY_INF, np.int64)
    if ng:
        samp[:] = key[np.minimum((np.arange(N_SAMPLES) * ng) // N_SAMPLES, ng - 1)]
    allsamp = yield ("all_gather", samp)            # [world, N_SAMPLES]
    flat = np.sort(np.asarray(allsamp).reshape(-1))
    flat = flat[flat != KEY_INF]
    cuts = []
    for j in range(1, world):
        target = int(flat[(j * len(flat)) // world]) if len(flat) else KEY_INF
        # ---- 3. move the cut forward to a global bundle boundary ------------------------------------------
        p = target
        for _ in range(10000):
            if p == KEY_INF:
                break
            i = int(np.searchsorted(key, p, side="left"))
            m_local = int(emax[i - 1]) if i > 0 else -1
            m = int((yield ("all_reduce_max", np.array([m_local], np.int64)))[0])
            if m < p:                               # every earlier group of every rank ends before p
                break
            i2 = int(np.searchsorted(key, m, side="right"))   # first local group starting after m
            nxt = int(key[i2]) if i2 < ng else KEY_INF
            p = int((yield ("all_reduce_min", np.array([nxt], np.int64)))[0])
        cuts.append(p)
    cuts = np.array(cuts, np.int64)
    dest = np.searchsorted(cuts, key, side="right") if world > 1 else np.zeros(ng, np.int64)

    # ---- 4. exchange the partial groups ---------------------------------------------------------------------
    ncig, coff, cig = _gather_cigars(local_tile, rep) if ng else (np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.uint32))
    prio_hi = np.asarray(loc["rep_effend"]).astype(np.int64)
    prio_lo = ((first_fidx + lf) << 32) | (rep - fo[lf]) if ng else np.zeros(0, np.int64)
    P = np.stack([tid, np.asarray(local_tile.pos)[rep].astype(np.int64) if ng else tid, np.asarray(local_tile.strand)[rep].astype(np.int64) if ng else tid,
                  np.asarray(loc["yx"]).astype(np.int64), np.asarray(loc["yd"]).astype(np.int64), prio_hi, prio_lo, ncig], axis=1) \
        if ng else np.zeros((0, 8), np.int64)
    yc = np.asarray(loc["yc"]).astype(np.float64)
    cnt = np.bincount(dest, minlength=world).astype(np.int64)
    ccnt = np.bincount(dest, weights=ncig, minlength=world).astype(np.int64) if ng else np.zeros(world, np.int64)
    rP, rcnt = yield ("all_to_all", (P, cnt))
    ryc, _ = yield ("all_to_all", (yc, cnt))
    rcig, rccnt = yield ("all_to_all", (cig.astype(np.int64), ccnt))

    # ---- 5. stitch: second-level collapse over the received partials ---------------------------------------
    n2 = int(rP.shape[0])
    file_off = np.zeros(world + 1, np.uint32)
    file_off[1:] = np.cumsum(rcnt)
    cig_off2 = np.zeros(n2 + 1, np.uint32)
    cig_off2[1:] = np.cumsum(rP[:, 7]) if n2 else 0
    tile2 = SoATile(
        n_files=world, file_off=file_off, tbmerged=np.ones(world, np.uint8), tid=rP[:, 0].astype(np.int32),
        pos=rP[:, 1].astype(np.int32), flag=np.zeros(n2, np.uint16), mapq=np.full(n2, 255, np.uint8),
        strand=rP[:, 2].astype(np.uint8), nh=np.full(n2, -(2**31), np.int32), cig_off=cig_off2,
        cig=np.asarray(rcig).astype(np.uint32), yc_in=np.asarray(ryc, np.float64), yx_in=rP[:, 3].copy(), yd_in=rP[:, 4].copy(),
        prio_hi=rP[:, 5].astype(np.uint64), prio_lo=rP[:, 6].astype(np.uint64))
    fin = compute.collapse(tile2, strategy=strategy, want_coords=True, keep_supplementary=True, keep_secondary=True)
    g2 = int(fin["n_groups"])
    rep2 = np.asarray(fin["rep"]).astype(np.int64)
    plo = rP[rep2, 6] if g2 else np.zeros(0, np.int64)
    cov_in = None
    ncg, cof, cg = _gather_cigars(tile2, rep2) if g2 else (np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.uint32))
    cov_in = CovInput(tid=tile2.tid[rep2] if g2 else np.zeros(0, np.int32), pos=tile2.pos[rep2] if g2 else np.zeros(0, np.int32),
                      flag=np.zeros(g2, np.uint16), cig_off=cof.astype(np.uint32), cig=cg,
                      yc=np.asarray(fin["yc"]).astype(np.float32).astype(np.float64),
                      strand=tile2.strand[rep2] if g2 else np.zeros(0, np.uint8), yx=np.asarray(fin["yx"]).astype(np.int64))
    res = ShardResult(n_groups=g2, n_passed_local=int(loc["n_passed"]), tid=cov_in.tid, start=np.asarray(fin["g_start"]),
                      end=np.asarray(fin["g_end"]), rep_fidx=(plo >> 32).astype(np.int64), rep_idx=(plo & 0xFFFFFFFF).astype(np.int64),
                      yc=np.asarray(fin["yc"]), yx=np.asarray(fin["yx"]), yd=np.asarray(fin["yd"]), cov_input=cov_in)
    # ---- 6. tiecov on the owned slice (whole bundles by construction of the cuts) ---------------------------
    if want_coverage:
        cov = compute.coverage(cov_in)
        nj = yield ("all_gather", np.array([int(cov["n_junctions"])], np.int64))
        res.coverage = cov
        res.junction_offset = int(np.asarray(nj).reshape(-1)[:rank].sum())
    return res


# ---- drivers ---------------------------------------------------------------------------------------------------
def run_loopback(compute, tiles, first_fidx, **kw):
    """Run R virtual ranks in one process: steps the R generators in lockstep and serves their collectives."""
    world = len(tiles)
    gens = [shard_collapse(compute, tiles[r], first_fidx[r], r, world, **kw) for r in range(world)]
    reqs = [next(g) for g in gens]
    results = [None] * world
    while any(r is None for r in results):
        kinds = {q[0] for q, res in zip(reqs, results) if res is None}
        assert len(kinds) == 1 and all(res is None for res in results), "ranks diverged"
        kind = kinds.pop()
        pay = [q[1] for q in reqs]
        if kind == "all_gather":
            out = [np.stack(pay)] * world
        elif kind == "all_reduce_max":
            out = [np.max(np.stack(pay), axis=0)] * world
        elif kind == "all_reduce_min":
            out = [np.min(np.stack(pay), axis=0)] * world
        elif kind == "all_to_all":
            out = []
            for d in range(world):
                parts, cnts = [], []
                for s in range(world):
                    data, cnt = pay[s]
                    o = int(cnt[:d].sum())
                    parts.append(data[o:o + int(cnt[d])])
                    cnts.append(int(cnt[d]))
                out.append((np.concatenate(parts), np.array(cnts, np.int64)))
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
    """One process per GPU: serve the generator's collectives with torch.distributed (RCCL on ROCm)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    gen = shard_collapse(compute, tile, first_fidx, rank, world, **kw)
    try:
        req = next(gen)
        while True:
            kind, pay = req
            if kind == "all_gather":
                x = t(pay)
                outs = [torch.empty_like(x) for _ in range(world)]
                dist.all_gather(outs, x, group=group)
                res = torch.stack(outs).cpu().numpy()
            elif kind in ("all_reduce_max", "all_reduce_min"):
                x = t(pay)
                dist.all_reduce(x, op=dist.ReduceOp.MAX if kind.endswith("max") else dist.ReduceOp.MIN, group=group)
                res = x.cpu().numpy()
            elif kind == "all_to_all":
                data, cnt = pay
                c = t(cnt)
                rc = torch.empty_like(c)
                dist.all_to_all_single(rc, c, group=group)          # who sends me how many rows
                rcnt = rc.cpu().numpy().astype(np.int64)
                x = t(data)
                out = torch.empty((int(rcnt.sum()),) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
                dist.all_to_all_single(out, x, output_split_sizes=rcnt.tolist(), input_split_sizes=np.asarray(cnt).tolist(),
                                       group=group)
                res = (out.cpu().numpy(), rcnt)
            else:
                raise AssertionError(kind)
            req = gen.send(res)
    except StopIteration as e:
        return e.value
