"""Device-side tagging + BGZF encode of the output records (tbk_bam_encode, bgzdef.hip) against the host writer path
(libtbh.so: tagwrite.cpp, the code the `tiebrush` command line tags with — flushPData's rules, /root/reference/src/tiebrush.cpp:
506-525, through htslib's bam_aux_update_* semantics): the inflated record stream must be byte-identical, every member must begin
with a record, carry the right CRC32 / ISIZE and inflate with zlib."""
import ctypes as C
import gzip
import os
import struct

import numpy as np
import pytest

from helpers import GOLDEN, sample_paths
from test_gpu_deflate import check_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def host_tagged_stream(records, yc, yx, yd, tmp_path):
    """the host writer's record stream: records (bytes without block_size) tagged by tbh_tag_deflate_part, inflated again"""
    from tiebrush_amd import _lib
    H = _lib.load_host()
    n = len(records)
    off = np.zeros(n + 1, dtype=np.uint64)
    ln = np.array([len(r) for r in records], dtype=np.uint32)
    off[1:] = np.cumsum(ln.astype(np.uint64))
    blob = np.frombuffer(b"".join(records) + b"\0", dtype=np.uint8)
    ycv = np.ascontiguousarray(yc, np.float64)
    yxv = np.ascontiguousarray(yx, np.int64)
    ydv = np.ascontiguousarray(yd, np.int32)
    path = str(tmp_path / "host_part.bgzf")
    rc = H.tbh_tag_deflate_part(blob.ctypes.data, off.ctypes.data, ln.ctypes.data, n, ycv.ctypes.data, yxv.ctypes.data, ydv.ctypes.data, 6, 2,
                                path.encode())
    assert rc == 0, H.tbh_last_error()
    data = open(path, "rb").read()
    return gzip.decompress(data) if data else b""


def members_begin_with_records(run: bytes):
    """every member's payload is a whole number of records (block_size chains end exactly at the member's end)"""
    from test_gpu_deflate import members
    import zlib
    for z, _, isize in members(run):
        p = zlib.decompress(z, -15)
        o = 0
        while o < len(p):
            o += 4 + struct.unpack_from("<I", p, o)[0]
        assert o == len(p) == isize


def _golden_case(ctx, names, tbmerged, **opts):
    files = [open(os.path.join(GOLDEN, n), "rb").read() for n in names]
    s, fo = ctx.bam_decode(files, tbmerged=np.array(tbmerged, np.uint8))
    g = ctx.collapse_struct(s, len(files), **opts)
    rep = g["rep"].cpu().numpy().astype(np.uint32)
    return s, rep, g["yc"].cpu().numpy(), g["yx"].cpu().numpy(), g["yd"].cpu().numpy()


@pytest.mark.parametrize("case", ["t1", "t2", "t12"])
def test_device_records_equal_the_host_writer(ctx, tmp_path, case):
    if case == "t12":
        names, tb = ["t1/t1.bam", "t2/t2.bam"], [1, 1]            # TieBrush-merged inputs: every record carries YC / YX (/ YD) already
    else:
        names, tb = [os.path.relpath(p, GOLDEN) for p in sample_paths(case)], [0] * 10
    s, rep, yc, yx, yd = _golden_case(ctx, names, tb)
    n_dev = int(s.n_records)
    run, pay = ctx.bam_encode(rep, yc, yx, yd, n_dev=n_dev)
    blob, off = ctx.bam_records(rep)
    recs = [blob[int(off[i]) + 4:int(off[i + 1])] for i in range(len(rep))]
    want = host_tagged_stream(recs, yc, yx, yd, tmp_path)
    assert pay == len(want)
    check_run(run, want)
    members_begin_with_records(run)
    # the same records handed over by the host instead (n_dev = 0), and half / half
    run2, _ = ctx.bam_encode(rep, yc, yx, yd, n_dev=0, host_records={i: recs[i] for i in range(len(rep))})
    assert gzip.decompress(run2) == want
    half = n_dev // 2
    run3, _ = ctx.bam_encode(rep, yc, yx, yd, n_dev=half, host_records={i: recs[i] for i in range(len(rep)) if rep[i] >= half})
    assert gzip.decompress(run3) == want
    ctx.bam_release()


def test_tag_edit_rules(ctx, tmp_path):
    """records that already carry the tags in every form bam_aux_update_* distinguishes (htslib 1.18 through GSam.h:300-305):
    YC as f / d / a wrong type, YX in every integer width, as a string, YD present with a zero and a positive result, duplicates,
    values at the width boundaries 254 / 255 / 65534 / 65535"""
    from tiebrush_amd import bamio
    auxes = [
        b"",
        b"NHC\x01",
        b"YCf" + struct.pack("<f", 3.0),
        b"YCd" + struct.pack("<d", 3.0) + b"NHC\x01",
        b"YCi" + struct.pack("<i", 3),
        b"XSA+" + b"YXC\x07" + b"ZZZhello\0",
        b"YXS" + struct.pack("<H", 700) + b"YCf" + struct.pack("<f", 1.0),
        b"YXI" + struct.pack("<I", 70000),
        b"YXc" + struct.pack("<b", -3),
        b"YXZabc\0",
        b"YDC\x05",
        b"YDI" + struct.pack("<I", 5) + b"NHC\x02",
        b"YDZxx\0",
        b"YXC\x01YXC\x02YDC\x09YDC\x08YCf" + struct.pack("<f", 2.0) + b"YCf" + struct.pack("<f", 9.0),
        b"BBBC" + struct.pack("<I", 3) + b"\x01\x02\x03" + b"YDs" + struct.pack("<h", 300),
        b"MDZ100\0YTZUU\0",
    ]
    vals = [(1.0, 1, 0), (2.0, 254, 0), (3.0, 255, 7), (70000.0, 65534, 254), (1e9, 65535, 255), (5.0, 70000, 65535), (2.5, 3, 65534), (7.0, 2 ** 32 + 5, 1),
            (4.0, 4, 70000)]
    recs, yc, yx, yd = [], [], [], []
    for a in auxes:
        for (c, x, d) in vals:
            r = bamio.encode_record(0, 100 + len(recs), 0, 60, [100 << 4], b"r%d" % len(recs), aux=a, l_seq=4, seq=b"\x12\x48", qual=b"IIII")
            recs.append(r[4:])
            yc.append(c), yx.append(x), yd.append(d)
    n = len(recs)
    rep = np.arange(n, dtype=np.uint32)
    run, pay = ctx.bam_encode(rep, yc, yx, yd, n_dev=0, host_records={i: recs[i] for i in range(n)})
    want = host_tagged_stream(recs, yc, yx, yd, tmp_path)
    got = gzip.decompress(run)
    if got != want:                      # say which record differs
        o = 0
        for i in range(n):
            bs = struct.unpack_from("<I", want, o)[0]
            assert got[o:o + 4 + bs] == want[o:o + 4 + bs], (i, auxes[i // len(vals)], vals[i % len(vals)])
            o += 4 + bs
    assert got == want and pay == len(want)
    members_begin_with_records(run)


def test_many_records_many_members(ctx, tmp_path):
    """a few hundred thousand output records: members cut at record boundaries by bisection, all of them exact"""
    from tiebrush_amd import bamio
    rng = np.random.default_rng(23)
    base = [bamio.encode_record(0, 1000 + k, 16 * (k & 1), 60, [100 << 4], b"read_%d" % k, aux=b"NHC\x01" + b"ASC" + bytes([k & 63]), l_seq=100,
                                seq=bytes(rng.integers(0, 256, 50, dtype=np.uint8)), qual=bytes(rng.integers(30, 42, 100, dtype=np.uint8)))[4:] for k in range(4000)]
    n = 300_000
    pick = rng.integers(0, len(base), n)
    recs = [base[i] for i in pick]
    yc = rng.integers(1, 400, n).astype(np.float64)
    yx = rng.integers(1, 70, n)
    yd = rng.integers(0, 3, n) * rng.integers(0, 70000, n)
    rep = np.arange(n, dtype=np.uint32)
    run, pay = ctx.bam_encode(rep, yc, yx, yd, n_dev=0, host_records={i: recs[i] for i in range(n)})
    want = host_tagged_stream(recs, yc, yx, yd, tmp_path)
    assert pay == len(want)
    check_run(run, want)
    members_begin_with_records(run)


def test_empty_and_malformed(ctx):
    run, pay = ctx.bam_encode(np.zeros(0, np.uint32), [], [], [])
    assert run == b"" and pay == 0
    from tiebrush_amd.api import TbkError
    bad = b"\x00" * 20                                   # shorter than a BAM core
    with pytest.raises(TbkError):
        ctx.bam_encode(np.zeros(1, np.uint32), [1.0], [1], [0], n_dev=0, host_records={0: bad})


def test_bad_split_arguments_are_refused_not_faulted(ctx):
    """tbk_bam_encode trusts neither side of the rep / n_dev / n_host split: a representative beyond n_dev with no host record behind it
    (no blob at all, or a slot beyond n_host) is TBK_EINVAL — it used to be a null / out-of-range dereference on the device"""
    import ctypes as C

    from tiebrush_amd import _lib
    from tiebrush_amd.api import TbkError
    # (a) rep >= n_dev, n_host == 0: nothing to read the record from
    with pytest.raises(TbkError) as e:
        ctx.bam_encode(np.array([5], np.uint32), [1.0], [1], [0], n_dev=0, host_records=None)
    assert e.value.status == -1                              # TBK_EINVAL
    # (b) a host slot beyond n_host
    from tiebrush_amd import bamio
    r0 = bamio.encode_record(0, 10, 0, 60, [(20 << 4) | 0], b"q", l_seq=0)[4:]
    rep = np.array([0, 1], np.uint32)
    en = _lib.EncIn()
    yc, yx, yd = np.ones(2), np.ones(2, np.int64), np.zeros(2, np.int32)
    slot = np.array([0, 7], np.uint32)                       # group 1 points at slot 7 of a blob that holds one record
    blob = np.frombuffer(len(r0).to_bytes(4, "little") + r0, dtype=np.uint8)
    off = np.array([0, 4 + len(r0)], np.uint64)
    en.mem, en.n, en.rep, en.yc, en.yx, en.yd, en.n_dev = _lib.TBK_MEM_HOST, 2, rep.ctypes.data, yc.ctypes.data, yx.ctypes.data, yd.ctypes.data, 0
    en.n_host, en.host_blob, en.host_off, en.host_slot = 1, blob.ctypes.data, off.ctypes.data, slot.ctypes.data
    out = np.empty(1 << 16, np.uint8)
    need, pay = C.c_uint64(0), C.c_uint64(0)
    rc = ctx.L.tbk_bam_encode(ctx.h, C.byref(en), out.ctypes.data, out.size, C.byref(need), C.byref(pay))
    assert rc == -1, rc                                       # TBK_EINVAL
    # the context is still usable
    run, _ = ctx.bam_encode(np.array([0], np.uint32), [1.0], [1], [0], n_dev=0, host_records={0: r0})
    assert len(run) > 0


@pytest.mark.parametrize("case", ["t1", "t12"])
def test_kept_results_feed_the_encoder(ctx, case):
    """tbk_collapse_opts.keep_results (ABI 8): the context keeps the call's rep / yc / yx / yd on the device; tbk_bam_encode with
    TBK_MEM_KEPT reads any range of them there and writes the bytes it writes from the caller's arrays; tbk_kept_results hands them out"""
    from tiebrush_amd.api import TbkError
    if case == "t12":
        names, tb = ["t1/t1.bam", "t2/t2.bam"], [1, 1]
    else:
        names, tb = [os.path.relpath(p, GOLDEN) for p in sample_paths(case)], [0] * 10
    s, rep, yc, yx, yd = _golden_case(ctx, names, tb, keep_results=True)
    m, n_dev = len(rep), int(s.n_records)
    assert m > 40
    want, pay = ctx.bam_encode(rep, yc, yx, yd, n_dev=n_dev)
    got, pay_k = ctx.bam_encode(rep, None, None, None, n_dev=n_dev, kept_first=0)
    assert got == want and pay_k == pay
    a, b = 7, m - 11                                            # a range in the middle: the writer's later chunks
    want_mid, _ = ctx.bam_encode(rep[a:b], yc[a:b], yx[a:b], yd[a:b], n_dev=n_dev)
    got_mid, _ = ctx.bam_encode(rep[a:b], None, None, None, n_dev=n_dev, kept_first=a)
    assert got_mid == want_mid
    # another context reads them (and the decoded tile) where they lie: tbk_enc_in.from — the command line's second encode thread
    from tiebrush_amd import api
    ctx2 = api.Context(0)
    try:
        got2, _ = ctx2.bam_encode(rep[a:b], None, None, None, n_dev=n_dev, kept_first=a, from_ctx=ctx)
        assert got2 == want_mid
        got3, _ = ctx2.bam_encode(rep, yc, yx, yd, n_dev=n_dev, from_ctx=ctx)        # (the caller's arrays, the other context's tile)
        assert got3 == want
        with pytest.raises(TbkError):                                               # ... which ctx2 itself does not have
            ctx2.bam_encode(rep, yc, yx, yd, n_dev=n_dev)
    finally:
        ctx2.close()
    # host records beside kept tag values (the command line's whole-input host path: rep on the host, the tags on the device)
    blob, off = ctx.bam_records(rep)
    recs = {i: blob[int(off[i]) + 4:int(off[i + 1])] for i in range(m)}
    got_host, _ = ctx.bam_encode(rep, None, None, None, n_dev=0, host_records=recs, kept_first=0)
    assert gzip.decompress(got_host) == gzip.decompress(want)
    r2, c2, x2, d2 = ctx.kept_results(0, m)
    assert np.array_equal(r2, rep) and np.array_equal(c2, yc) and np.array_equal(x2, yx) and np.array_equal(d2, yd)
    r3, c3, x3, d3 = ctx.kept_results(a, b - a, tags_only=True)
    assert r3 is None and np.array_equal(c3, yc[a:b]) and np.array_equal(x3, yx[a:b]) and np.array_equal(d3, yd[a:b])
    with pytest.raises(TbkError):                               # a range that ends behind the kept groups
        ctx.bam_encode(rep[:5], None, None, None, n_dev=n_dev, kept_first=m - 4)
    with pytest.raises(TbkError):
        ctx.kept_results(m - 1, 2)
    # the next collapse without the option leaves nothing behind
    g = ctx.collapse_struct(s, len(names))
    assert g["n_groups"] == m
    with pytest.raises(TbkError):
        ctx.bam_encode(rep[:5], None, None, None, n_dev=n_dev, kept_first=0)
    with pytest.raises(TbkError):
        ctx.kept_results(0, 1)
    with pytest.raises(TbkError):                               # the YD column of a deferred stage is not final when the call returns
        ctx.collapse_struct(s, len(names), keep_results=True, defer_yd=True)
    ctx.bam_release()


def test_keep_results_lets_a_host_caller_leave_the_tag_arrays_out(ctx):
    """with keep_results a TBK_MEM_HOST caller hands over `rep` only (what the command line's whole-input paths do): n_groups / rep as
    ever, the values on the device equal to those of an ordinary call; without the option NULL tag arrays stay TBK_EINVAL"""
    from tiebrush_amd import _lib, synth
    tile = synth.make_tile(5, 30000, "c2", n_loci=300)
    ref = ctx.collapse(tile)
    m = ref["n_groups"]
    keep = []
    s, dev, n = ctx._soa_struct(tile, keep)
    assert not dev
    rep = np.empty(n, np.uint32)
    want_rep = np.asarray(ref["rep"]).view(np.uint32)
    for keep_results, with_rep, want_rc in ((1, 1, 0), (1, 0, 0), (0, 1, -1)):
        o = ctx.make_opts(keep_results=bool(keep_results))
        rep[:] = 0xFFFFFFFF
        g = _lib.GroupsOut(_lib.TBK_MEM_HOST, n, rep.ctypes.data if with_rep else None, None, None, None, None, None, None, None, None, 0, 0)
        assert ctx.L.tbk_collapse_tile(ctx.h, C.byref(o), C.byref(s), C.byref(g)) == want_rc
        if want_rc == 0:
            assert int(g.n_groups) == m
            if with_rep:
                assert np.array_equal(rep[:m], want_rep)
            r2, c2, x2, d2 = ctx.kept_results(0, m)        # (no array at all: `rep` too comes once n_groups is known — the command line's way)
            assert np.array_equal(r2, want_rep) and np.array_equal(c2, ref["yc"]) and np.array_equal(x2, ref["yx"]) and np.array_equal(d2, ref["yd"])


def test_results_of_moderate_size_come_back_through_the_staging_buffer(ctx):
    """results between 1 and 64 MB an array take the context's page-locked staging buffer and a copy by the core instead of a
    registration of the caller's array (tbk_api.hip: d2h); TBK_DEBUG no_bounce=1 is the direct way.  Same bytes, any size around the
    buffer's 8 MB; tbk_kept_results goes the same way; tbk_warmup (the command line's helper thread calls it) changes nothing"""
    from tiebrush_amd import synth
    tile = synth.make_tile(8, 400000, "c2", n_loci=200000)
    a = ctx.collapse(tile, want_rec_group=True)
    m = a["n_groups"]
    assert m * 8 > (8 << 20) and m * 4 > (1 << 20)              # yc / yx need two passes of the buffer, rep / yd one
    assert ctx.L.tbk_warmup(ctx.h) == 0
    ctx.L.tbk_set_debug(ctx.h, b"no_bounce=1")
    try:
        b = ctx.collapse(tile, want_rec_group=True)
    finally:
        ctx.L.tbk_set_debug(ctx.h, b"")
    for k in ("rep", "yc", "yx", "yd", "g_start", "g_end", "rec_group"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    c = ctx.collapse(tile, keep_results=True)
    r, yc, yx, yd = ctx.kept_results(0, m)
    assert np.array_equal(r, np.asarray(a["rep"]).view(np.uint32)) and np.array_equal(yc, a["yc"]) and np.array_equal(yx, a["yx"]) and np.array_equal(yd, a["yd"])
    r, yc, yx, yd = ctx.kept_results(m // 3, m - m // 3 - 5)
    assert np.array_equal(yx, a["yx"][m // 3:m - 5]) and np.array_equal(r, np.asarray(a["rep"]).view(np.uint32)[m // 3:m - 5])
