"""Oracle-backed `compute` object for the CPU tests of tiebrush_amd.dist (test infrastructure: the product's
compute object is tiebrush_amd.api.Context).  Only the per-tile compute is substituted; the sharding /
exchange / stitch logic under test is the product's."""
import numpy as np

from oracle import oracle_ffi as orc

STRAT = {"cigar": 0, "full": 1, "clip": 2, "exon": 3}


def reflen(tile):
    ops = tile.cig & 0xF
    ln = (tile.cig >> 4).astype(np.int64)
    w = np.where(np.isin(ops, [0, 2, 3, 7, 8]), ln, 0)
    c = np.concatenate([[0], np.cumsum(w)])
    co = tile.cig_off.astype(np.int64)
    return c[co[1:]] - c[co[:-1]]


def effend_all(tile):
    """per-file running max of `end` inside runs of equal (tid,start): the merge key of tmerge.h:28-50"""
    n = tile.n_records
    end = tile.pos.astype(np.int64) + reflen(tile)
    f = tile.file_of().astype(np.int64)
    key = (f << 48) | ((tile.tid.astype(np.int64) + 1) << 32) | (tile.pos.astype(np.int64) + 1)
    head = np.ones(n, bool)
    head[1:] = key[1:] != key[:-1]
    run = np.cumsum(head) - 1
    big = np.int64(1) << 33
    return (np.maximum.accumulate(run * big + end) - run * big).astype(np.int64)


class OracleCompute:
    def collapse(self, tile, strategy="cigar", want_coords=True, want_effend=False, want_rec_group=False, **kw):
        r = orc.collapse(tile, strategy=STRAT[strategy], want_rec_group=True, **kw)
        if tile.prio_hi is not None and r["n_groups"]:
            rg = r["rec_group"]
            ok = rg >= 0
            idx = np.nonzero(ok)[0]
            order = np.lexsort((tile.prio_lo[idx].astype(np.int64), tile.prio_hi[idx].astype(np.int64), rg[idx]))
            srt = idx[order]
            first = np.ones(len(srt), bool)
            first[1:] = rg[srt][1:] != rg[srt][:-1]
            r["rep"] = srt[first].astype(np.uint32)
        if want_effend:
            r["rep_effend"] = effend_all(tile)[r["rep"].astype(np.int64)].astype(np.int32) if r["n_groups"] else np.zeros(0, np.int32)
        return r

    def coverage(self, cin):
        return orc.coverage(cin)


def split_tile(tile, parts):
    """Cut a file-major tile into per-rank tiles of consecutive files; returns (tiles, first_fidx)."""
    from tiebrush_amd.soa import SoATile
    k = tile.n_files
    bounds = [(k * r) // parts for r in range(parts + 1)]
    tiles, first = [], []
    for r in range(parts):
        f0, f1 = bounds[r], bounds[r + 1]
        lo, hi = int(tile.file_off[f0]), int(tile.file_off[f1])
        c0, c1 = int(tile.cig_off[lo]), int(tile.cig_off[hi])
        t = SoATile(n_files=f1 - f0, file_off=(tile.file_off[f0:f1 + 1] - tile.file_off[f0]).astype(np.uint32),
                    tbmerged=tile.tbmerged[f0:f1].copy(), tid=tile.tid[lo:hi].copy(), pos=tile.pos[lo:hi].copy(),
                    flag=tile.flag[lo:hi].copy(), mapq=tile.mapq[lo:hi].copy(), strand=tile.strand[lo:hi].copy(),
                    nh=tile.nh[lo:hi].copy(), cig_off=(tile.cig_off[lo:hi + 1] - tile.cig_off[lo]).astype(np.uint32),
                    cig=tile.cig[c0:c1].copy())
        if tile.yc_in is not None:
            t.yc_in, t.yx_in, t.yd_in = tile.yc_in[lo:hi].copy(), tile.yx_in[lo:hi].copy(), tile.yd_in[lo:hi].copy()
        if tile.md_off is not None:       # -L: the MD strings of the rank's records
            m0, m1 = int(tile.md_off[lo]), int(tile.md_off[hi])
            t.md_off, t.md, t.md_has = (tile.md_off[lo:hi + 1] - tile.md_off[lo]).astype(np.uint32), tile.md[m0:m1].copy(), tile.md_has[lo:hi].copy()
        tiles.append(t)
        first.append(f0)
    return tiles, first


def check_against_flat(results, tile, flat, flat_cov=None):
    """Concatenated shard results == the flat (single tile) run: same groups in the same order, same YC/YX/YD and the
    same representative record; optionally the same bedgraph intervals / junction rows."""
    cat = lambda name: np.concatenate([np.asarray(getattr(r, name)) for r in results])
    assert sum(r.n_groups for r in results) == flat["n_groups"]
    assert sum(r.n_passed_local for r in results) == flat["n_passed"]
    rep = flat["rep"].astype(np.int64)
    fo = tile.file_of().astype(np.int64)
    assert np.array_equal(cat("start"), flat["g_start"]) and np.array_equal(cat("end"), flat["g_end"])
    assert np.array_equal(cat("tid"), tile.tid[rep])
    assert np.array_equal(cat("yc"), flat["yc"]) and np.array_equal(cat("yx"), flat["yx"]) and np.array_equal(cat("yd"), flat["yd"])
    assert np.array_equal(cat("rep_fidx"), fo[rep])
    assert np.array_equal(cat("rep_idx"), rep - tile.file_off[fo[rep]].astype(np.int64))
    if flat_cov is not None:
        for k in ("iv_tid", "iv_start", "iv_end", "iv_val", "j_tid", "j_start", "j_end", "j_strand", "j_val"):
            got = np.concatenate([np.asarray(r.coverage[k]) for r in results])
            assert np.array_equal(got, flat_cov[k]), k
        offs = [r.junction_offset for r in results]
        assert offs == list(np.concatenate([[0], np.cumsum([r.coverage["n_junctions"] for r in results])])[:-1])
