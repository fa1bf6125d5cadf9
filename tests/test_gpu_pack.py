"""The packed wire form of a tile (tbk_packed_in / tbk_unpack_tile, ABI 5): the device rebuilds the structure of arrays bit for bit,
and a collapse of the unpacked tile equals the collapse of the plain tile and the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FIELDS = ("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig")
KEYS = ("rep", "yc", "yx", "yd", "g_start", "g_end")


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _roundtrip(ctx, tile, **kw):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, soa
    pt = soa.pack_tile(tile)
    dt = ctx.unpack_tile(pt)
    got = ctx.soa_to_numpy(dt.struct, fields=FIELDS)
    for f in FIELDS:
        assert np.array_equal(got[f], np.asarray(getattr(tile, f))), f
    okw = dict(kw)
    if "strategy" in okw:
        okw["strategy"] = {"cigar": 0, "clip": 2, "exon": 3}[okw["strategy"]]
    want = orc.collapse(tile, **okw)
    a = api.to_numpy(ctx.collapse(dt, **kw))
    b = api.to_numpy(ctx.collapse(api.to_device(tile, "cuda:0"), **kw))
    for r in (a, b):
        assert r["n_groups"] == want["n_groups"] and r["n_passed"] == want["n_passed"]
        for k in KEYS:
            assert np.array_equal(np.asarray(r[k]), np.asarray(want[k])), k
    return pt


@pytest.mark.parametrize("profile,kw", [("c2", {}), ("c3", dict(strategy="clip")), ("c5", dict(strategy="exon", max_nh=5, min_qual=1))])
@pytest.mark.parametrize("n_files,reads", [(5, 30000), (70, 3000)])
def test_unpacked_tile_equals_the_plain_tile(ctx, profile, kw, n_files, reads):
    from tiebrush_amd import synth
    tile = synth.make_tile(n_files=n_files, reads_per_file=reads, profile=profile, n_loci=500)
    pt = _roundtrip(ctx, tile, **kw)
    plain = sum(int(np.asarray(getattr(tile, f)).nbytes) for f in FIELDS)
    assert pt.nbytes() < 0.7 * plain                                 # 9 bytes + CIGAR words instead of 20 + CIGAR words per record


def test_escapes_and_edges(ctx):
    """NH beyond the 10-bit code and absent, a CIGAR of 300 operations, a file that is empty, reference changes inside a file, a
    window-path sized tile (the unpacked tile is a device tile like any other)"""
    from tiebrush_amd import soa, synth
    tile = synth.make_tile(n_files=3, reads_per_file=5000, profile="c5", n_loci=100)
    tile.nh = tile.nh.copy()
    tile.nh[7] = 5000
    tile.nh[8] = soa.NH_ABSENT
    tile.nh[9] = 1021
    tile.nh[10] = 1022
    # a long CIGAR on one record: 150 x (1M 1I) keeps the reference length small
    i = 100
    co = tile.cig_off.astype(np.int64)
    new = np.array([(1 << 4) | (k & 1) for k in range(300)], np.uint32)
    tile.cig = np.concatenate([tile.cig[:co[i]], new, tile.cig[co[i + 1]:]])
    delta = 300 - int(co[i + 1] - co[i])
    tile.cig_off = tile.cig_off.copy()
    tile.cig_off[i + 1:] = (co[i + 1:] + delta).astype(np.uint32)
    _roundtrip(ctx, tile, keep_secondary=True)
    # an empty file in the middle
    e = synth.make_tile(n_files=3, reads_per_file=2000, profile="c2", n_loci=50)
    fo = e.file_off.copy()
    n1 = int(fo[2] - fo[1])
    keep = np.ones(e.n_records, bool)
    keep[int(fo[1]):int(fo[2])] = False
    cnt = np.diff(e.cig_off.astype(np.int64))
    rec_of = np.repeat(np.arange(e.n_records), cnt)
    e2 = soa.SoATile(n_files=3, file_off=np.array([0, fo[1], fo[1], fo[3] - n1], np.uint32), tbmerged=e.tbmerged, tid=e.tid[keep], pos=e.pos[keep],
                     flag=e.flag[keep], mapq=e.mapq[keep], strand=e.strand[keep], nh=e.nh[keep],
                     cig_off=np.concatenate([[0], np.cumsum(cnt[keep])]).astype(np.uint32), cig=e.cig[keep[rec_of]])
    _roundtrip(ctx, e2)


def test_packer_refuses_what_the_wire_form_does_not_carry(bam_loader):
    import os
    from helpers import GOLDEN
    from tiebrush_amd import soa, synth
    t = soa.tile_from_bams([bam_loader(os.path.join(GOLDEN, "t1", "t1.bam"))])
    with pytest.raises(ValueError):
        soa.pack_tile(t)                                             # TieBrush-merged: carried YC / YX / YD
    t2 = synth.make_tile(2, 100, "c2", n_loci=10)
    t2.flag = t2.flag.copy()
    t2.flag[3] |= 0x4000
    with pytest.raises(ValueError):
        soa.pack_tile(t2)
