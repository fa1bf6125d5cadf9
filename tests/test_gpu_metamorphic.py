"""Metamorphic identities on the HIP path (tests/metamorphic.py): `-P` == default on the soft-clip-stripped tile, `-E` forms the
default groups on an M/N-only tile, `-N / -Q` == default with the dropped records kept as merge-order ballast — on c3- / c5-profile
tiles, through the window path and the sort path, each compared with the HIP path's own default mode AND with the oracle."""
import numpy as np
import pytest

from helpers import tbk_debug

import metamorphic as mm

pytestmark = pytest.mark.gpu
KEYS = ("rep", "yc", "yx", "yd", "g_start", "g_end")


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _hip(ctx, tile, **kw):
    from tiebrush_amd import api
    return api.to_numpy(ctx.collapse(api.to_device(tile, "cuda:0"), **kw))


def _same(a, b):
    assert a["n_groups"] == b["n_groups"] and a["n_passed"] == b["n_passed"]
    for k in KEYS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def _paths(monkeypatch, path):
    tbk_debug(monkeypatch, path=None)
    if path != "auto":
        tbk_debug(monkeypatch, path=path)


@pytest.mark.parametrize("path", ["auto", "window", "sort"])
@pytest.mark.parametrize("n_files,reads", [(6, 40000), (70, 4000)])
def test_clip_equals_default_on_the_stripped_tile(ctx, monkeypatch, path, n_files, reads):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth
    _paths(monkeypatch, path)
    tile = synth.make_tile(n_files=n_files, reads_per_file=reads, profile="c3", n_loci=400)
    stripped = mm.strip_soft_clips(tile)
    clip = _hip(ctx, tile, strategy="clip")
    _same(clip, _hip(ctx, stripped, strategy="cigar"))
    _same(clip, orc.collapse(stripped, strategy=0))
    _same(clip, orc.collapse(tile, strategy=2))


@pytest.mark.parametrize("path", ["auto", "window", "sort"])
@pytest.mark.parametrize("profile", ["c3", "c5"])
def test_exon_forms_the_default_groups_on_an_MN_only_tile(ctx, monkeypatch, path, profile):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth
    _paths(monkeypatch, path)
    tile = synth.make_tile(n_files=5, reads_per_file=40000, profile=profile, n_loci=400)
    t = mm.drop_indel_reads(tile)
    if profile == "c3":
        t = mm.strip_soft_clips(t)
    assert mm.has_only_MN(t)
    kw = dict(keep_secondary=True, keep_supplementary=True)
    ex = _hip(ctx, t, strategy="exon", **kw)
    mm.same_groups_any_tie_order(ex, _hip(ctx, t, strategy="cigar", **kw))
    mm.same_groups_any_tie_order(ex, orc.collapse(t, strategy=0, **kw))
    _same(ex, orc.collapse(t, strategy=3, **kw))


@pytest.mark.parametrize("path", ["auto", "window", "sort"])
@pytest.mark.parametrize("strategy", ["cigar", "exon"])
@pytest.mark.parametrize("n_files,reads", [(7, 30000), (130, 2000)])
def test_nh_and_mapq_filters_equal_default_with_ballast(ctx, monkeypatch, path, strategy, n_files, reads):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth
    _paths(monkeypatch, path)
    tile = synth.make_tile(n_files=n_files, reads_per_file=reads, profile="c5", n_loci=400)
    t2, dropped = mm.ballast(tile, 5, 1)
    assert dropped > 1000
    s = {"cigar": 0, "exon": 3}[strategy]
    a = _hip(ctx, tile, strategy=strategy, max_nh=5, min_qual=1)
    _same(a, _hip(ctx, t2, strategy=strategy))
    _same(a, orc.collapse(t2, strategy=s))
    _same(a, orc.collapse(tile, strategy=s, max_nh=5, min_qual=1))


def _dev_strip_soft_clips(tile):
    """strip_soft_clips on a device tile (torch)"""
    import torch
    from dataclasses import replace
    cig = tile.cig.to(torch.int64) & 0xFFFFFFFF
    keep = (cig & 0xF) != mm.S_OP
    co = tile.cig_off.to(torch.int64) & 0xFFFFFFFF
    ck = torch.cat([torch.zeros(1, dtype=torch.int64, device=cig.device), torch.cumsum(keep.to(torch.int64), 0)])
    new_off = ck[co]
    assert int((new_off[1:] - new_off[:-1]).min()) > 0
    return replace(tile, cig=tile.cig[keep].contiguous(), cig_off=new_off.to(torch.int32))


def _dev_same(a, b):
    import torch
    assert a["n_groups"] == b["n_groups"] and a["n_passed"] == b["n_passed"]
    for k in KEYS:
        assert torch.equal(a[k], b[k]), k


def test_clip_equals_default_on_the_stripped_tile_config3_shape_32M(ctx):
    """the same identity at a tenth of config 3 (64 files x 500 k reads, generated and stripped on the device): the window path at the
    size where it is the default, every output array"""
    from tiebrush_amd import synth_dev
    tile = synth_dev.make_tile_device(64, 500_000, "c3", device="cuda:0")
    clip = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in ctx.collapse(tile, strategy="clip").items() if not k.startswith("_")}
    stripped = _dev_strip_soft_clips(tile)
    _dev_same(clip, ctx.collapse(stripped, strategy="cigar"))
    plain = ctx.collapse(tile, strategy="cigar")
    assert plain["n_groups"] > clip["n_groups"]


def test_filters_equal_default_with_ballast_config5_shape_32M(ctx):
    """-N 5 -Q 1 --exon on config 5's per-rank shape at a quarter of its size (128 files x 250 k reads) == the default filters on the
    tile whose dropped records became secondary alignments with passing NH / MAPQ (merge-order ballast), on the device"""
    import torch
    from dataclasses import replace
    from tiebrush_amd import synth_dev
    tile = synth_dev.make_tile_device(128, 250_000, "c5", device="cuda:0")
    drop = (tile.nh > 5) | (tile.mapq.to(torch.int32) < 1)
    assert int(drop.sum()) > 100000
    t2 = replace(tile, flag=torch.where(drop, tile.flag | 0x100, tile.flag), nh=torch.where(drop, torch.ones_like(tile.nh), tile.nh),
                 mapq=torch.where(drop, torch.full_like(tile.mapq, 60), tile.mapq))
    a = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in ctx.collapse(tile, strategy="exon", max_nh=5, min_qual=1).items()
         if not k.startswith("_")}
    _dev_same(a, ctx.collapse(t2, strategy="exon"))
