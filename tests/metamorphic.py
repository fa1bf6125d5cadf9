"""Tile transformations behind the metamorphic parity tests: identities that follow from the reference's source alone
(tiebrush.cpp:312-345 compare functions, :532-541 filter, :573 order fixed before filtering), so they cross-check the oracle's
and the HIP path's strategy / filter code without trusting either.

  strip_soft_clips(T)   cmpCigarClip(a, b) == cmpCigar(strip a, strip b) (tiebrush.cpp:312-332 vs :304-310) and an S operation
                        moves neither pos nor end nor the exons (GSam.cpp:351-417): `-P` on T == default on strip(T), every array.
  ballast(T, N, Q)      a record that -N / -Q drops still takes part in the merge order (tmerge.cpp:331-344 runs before
                        passes_options, tiebrush.cpp:573).  Marked secondary (0x100, dropped by the default filter too) and given
                        a passing NH / MAPQ it is the same ballast under default options: `-N n -Q q` on T == default on ballast(T).
  has_only_MN(T)        on a tile whose CIGARs hold only M and N, exon lists and CIGARs determine each other: `-E` forms the groups
                        of the default mode (the order inside a (strand, end) tie may differ: cmpExons vs memcmp of the words).
"""
import copy

import numpy as np

S_OP = 4


def strip_soft_clips(tile):
    t = copy.copy(tile)
    keep = (tile.cig & 0xF) != S_OP
    n = tile.n_records
    rec_of = np.repeat(np.arange(n), np.diff(tile.cig_off.astype(np.int64)))
    cnt = np.bincount(rec_of[keep], minlength=n)
    assert cnt.min() > 0, "a CIGAR of soft clips only"
    t.cig = np.ascontiguousarray(tile.cig[keep])
    t.cig_off = np.zeros(n + 1, np.uint32)
    t.cig_off[1:] = np.cumsum(cnt)
    return t


def ballast(tile, max_nh, min_qual):
    from tiebrush_amd.soa import NH_ABSENT
    t = copy.copy(tile)
    nh = np.where(tile.nh == NH_ABSENT, 0, tile.nh)           # (NH absent counts as 0: tiebrush.cpp:539)
    drop = (nh > max_nh) | (tile.mapq.astype(np.int64) < min_qual)
    t.flag = np.where(drop, tile.flag | 0x100, tile.flag).astype(np.uint16)
    t.nh = np.where(drop, 1, tile.nh).astype(np.int32)
    t.mapq = np.where(drop, 60, tile.mapq).astype(np.uint8)
    return t, int(drop.sum())


def drop_indel_reads(tile):
    """the same tile with every read that carries an I or D operation removed from its file (file order kept)"""
    from tiebrush_amd.soa import SoATile
    n = tile.n_records
    co = tile.cig_off.astype(np.int64)
    rec_of = np.repeat(np.arange(n), np.diff(co))
    op = tile.cig & 0xF
    bad = np.bincount(rec_of[(op == 1) | (op == 2)], minlength=n) > 0
    keep = ~bad
    fo = np.zeros(tile.n_files + 1, np.uint32)
    f_of = tile.file_of()
    fo[1:] = np.cumsum(np.bincount(f_of[keep], minlength=tile.n_files))
    kc = keep[rec_of]
    cnt = np.diff(co)[keep]
    cig_off = np.zeros(int(keep.sum()) + 1, np.uint32)
    cig_off[1:] = np.cumsum(cnt)
    return SoATile(n_files=tile.n_files, file_off=fo, tbmerged=tile.tbmerged.copy(), tid=tile.tid[keep], pos=tile.pos[keep],
                   flag=tile.flag[keep], mapq=tile.mapq[keep], strand=tile.strand[keep], nh=tile.nh[keep], cig_off=cig_off,
                   cig=np.ascontiguousarray(tile.cig[kc]))


def has_only_MN(tile):
    op = tile.cig & 0xF
    return bool(np.all((op == 0) | (op == 3)))


def same_groups_any_tie_order(a, b):
    """two collapse results describe the same groups: equal (rep, yc, yx) sets; where the output order is the same too, YD must
    agree as well (inside a (strand, end) tie -E and the default mode may order the groups differently, and the YD list machine
    is order dependent).  Returns whether the order was the same."""
    assert a["n_groups"] == b["n_groups"] and a["n_passed"] == b["n_passed"]
    ra, rb = np.asarray(a["rep"]).astype(np.int64), np.asarray(b["rep"]).astype(np.int64)
    oa, ob = np.argsort(ra, kind="stable"), np.argsort(rb, kind="stable")
    assert np.array_equal(ra[oa], rb[ob])
    for k in ("yc", "yx", "g_start", "g_end"):
        assert np.array_equal(np.asarray(a[k])[oa], np.asarray(b[k])[ob]), k
    same = bool(np.array_equal(ra, rb))
    if same:
        assert np.array_equal(np.asarray(a["yd"]), np.asarray(b["yd"]))
    return same
