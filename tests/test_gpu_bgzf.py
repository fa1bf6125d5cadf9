"""Device-side BGZF inflate (tbk_bgzf_inflate, bamdev.hip) against zlib: stored, fixed-code and dynamic-code deflate
blocks, every compression level, the reference's own fixture files, and rejection of corrupt members."""
import os
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, sample_paths, tbk_debug

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _bgzf(payload: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, block=0xff00) -> bytes:
    out = []
    for o in range(0, max(len(payload), 1), block):
        chunk = payload[o:o + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        c = co.compress(chunk) + co.flush()
        bsize = len(c) + 25
        hdr = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, ord("B"), ord("C"), 2, 0, bsize & 0xff, bsize >> 8])
        out.append(hdr + c + (zlib.crc32(chunk) & 0xFFFFFFFF).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
    return b"".join(out)


def _payloads():
    rng = np.random.default_rng(3)
    text = (b"ACGTTGCA" * 40 + b"read_%d\tchr1\t100M\n") * 3000
    return {
        "random": rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),            # incompressible: stored / long codes
        "text": text,                                                                  # long matches, overlapping copies
        "runs": b"\0" * 70_000 + b"\xff" * 70_000 + bytes(range(256)) * 300,           # distance-1 copies
        "mixed": rng.integers(0, 4, 200_000, dtype=np.uint8).tobytes() + text[:100_000],
        # matches from 9 - 30 KB back: beyond the wave kernel's LDS ring (the bytes come back from memory), up to deflate's reach
        "far": b"".join(blk + rng.integers(0, 256, gap, dtype=np.uint8).tobytes() + blk
                        for blk, gap in ((rng.integers(0, 256, 700, dtype=np.uint8).tobytes(), g) for g in (8100, 8200, 9000, 15000, 31000, 20000))),
        "tiny": b"x",
        "empty_then_data": b"",
    }


@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY])
def test_inflate_matches_zlib(ctx, level, strategy):
    for name, p in _payloads().items():
        comp = _bgzf(p, level, strategy)
        if name == "empty_then_data":
            comp = comp + _bgzf(b"after an empty member", level, strategy)
            p = b"after an empty member"
        assert ctx.bgzf_inflate(comp) == p, (name, level, strategy)


def test_inflate_reference_fixtures(ctx):
    from tiebrush_amd import bamio
    for path in [os.path.join(GOLDEN, "t12.bam"), sample_paths("t1")[0], sample_paths("t2")[3]]:
        raw = open(path, "rb").read()
        assert ctx.bgzf_inflate(raw) == bamio.bgzf_decompress(raw)


def test_corrupt_members_are_refused(ctx):
    from tiebrush_amd import api
    good = _bgzf((b"tiebrush " * 5000), 6)
    for what in ("crc", "body", "isize", "bsize"):
        b = bytearray(good)
        bs = b[16] | (b[17] << 8)
        if what == "crc":
            b[bs + 1 - 8] ^= 0x55
        elif what == "body":
            b[40] ^= 0xFF
        elif what == "isize":
            b[bs + 1 - 4] ^= 0x01
        else:
            b[16:18] = (9).to_bytes(2, "little")
        with pytest.raises(api.TbkError):
            ctx.bgzf_inflate(bytes(b))
    assert ctx.bgzf_inflate(good) == b"tiebrush " * 5000


def test_mutated_streams_are_refused_or_exact(ctx):
    """bit flips and truncations inside the deflate stream: the CRC leaves two outcomes, the payload or a refusal"""
    from tiebrush_amd import api
    rng = np.random.default_rng(11)
    payload = (b"ACGTTGCA" * 7 + b"read\t%d\n") * 900 + rng.integers(0, 256, 9000, dtype=np.uint8).tobytes()
    for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (9, zlib.Z_HUFFMAN_ONLY), (0, zlib.Z_DEFAULT_STRATEGY)):
        good = _bgzf(payload, level, strategy)
        bs = (good[16] | (good[17] << 8)) + 1     # first member
        refused = 0
        for _ in range(40):
            b = bytearray(good)
            at = int(rng.integers(18, bs - 8))
            b[at] ^= 1 << int(rng.integers(0, 8))
            try:
                assert ctx.bgzf_inflate(bytes(b)) == payload
            except api.TbkError:
                refused += 1
        assert refused >= 30, (level, refused)
        # a member whose stream is cut short (BSIZE and the trailer moved up): refused
        cut = bytearray(good[:bs])
        for drop in (1, 7, 200):
            if bs - 26 - drop < 1:
                continue
            c = bytearray(cut[:18]) + cut[18:bs - 8 - drop] + cut[bs - 8:bs]
            c[16:18] = (len(c) - 1).to_bytes(2, "little")
            try:
                assert ctx.bgzf_inflate(bytes(c) + good[bs:]) == payload and drop < 8   # (up to 8 zero bytes may stand in for the tail)
            except api.TbkError:
                pass


def test_bam_decode_matches_host_decoders(ctx, bam_loader):
    """tbk_bam_decode (inflate + record index + field / aux scan on the GPU) == the Python decoder of the same files: every
    SoA array, MD and names included, carried tags of a TieBrush-merged input, and the raw records behind tile indices."""
    from tiebrush_amd import bamio, soa
    paths = [os.path.join(GOLDEN, "t1", "t1.bam")] + sample_paths("t2")[:3]
    raw = [open(p, "rb").read() for p in paths]
    bams = [bam_loader(p, keep_md=True) for p in paths]
    tile = soa.tile_from_bams(bams, with_names=True, with_md=True)
    s, fo = ctx.bam_decode(raw, tbmerged=tile.tbmerged, want_md=True, want_names=True)
    assert np.array_equal(fo, tile.file_off) and s.n_records == tile.n_records and s.n_cigar_ops == tile.cig.shape[0]
    got = ctx.soa_to_numpy(s, fields=("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig", "yc_in", "yx_in", "yd_in", "md_off",
                                      "md_has", "qname_hash", "qname_off"))
    for name in ("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig", "md_off", "md_has", "qname_hash"):
        assert np.array_equal(got[name], getattr(tile, name)), name
    assert np.array_equal(got["md"], tile.md) and np.array_equal(got["qname_off"], tile.qn_off) and np.array_equal(got["qname"], tile.qn)
    n0 = int(tile.file_off[1])                      # carried tags are only defined for the TieBrush-merged file
    assert np.array_equal(got["yc_in"][:n0], tile.yc_in[:n0]) and np.array_equal(got["yx_in"][:n0], tile.yx_in[:n0])
    assert np.array_equal(got["yd_in"][:n0], tile.yd_in[:n0])
    # raw records behind some indices, in the order asked
    idx = np.array([0, 5, n0 - 1, n0, tile.n_records - 1, 17], np.uint32)
    blob, off = ctx.bam_records(idx)
    inflated = [bamio.bgzf_decompress(r) for r in raw]
    fo64 = tile.file_off.astype(np.int64)
    for j, i in enumerate(idx.tolist()):
        f = int(np.searchsorted(fo64, i, side="right") - 1)
        _, p = bamio.parse_header(inflated[f])
        for _ in range(i - int(fo64[f])):
            p += 4 + int.from_bytes(inflated[f][p:p + 4], "little")
        bs = int.from_bytes(inflated[f][p:p + 4], "little")
        assert blob[int(off[j]):int(off[j + 1])] == inflated[f][p:p + 4 + bs]
    ctx.bam_release()


def test_record_index_by_member_and_by_chain(ctx, monkeypatch):
    """the record index of tbk_bam_decode: a lane per BGZF member where every member begins with a record (files written through
    htslib: the reference's fixtures), the chain of block_size fields per file otherwise (blocks cut at a fixed size: bamio's
    writer) or when the index_chain hook asks for it — the same tile every way, and the kernel that ran is the one expected"""
    from tiebrush_amd import bamio
    gold = [open(p, "rb").read() for p in sample_paths("t2")[:4] + [os.path.join(GOLDEN, "t12.bam")]]
    cut = []                                              # the same records in blocks of 0x1234 payload bytes: records span blocks
    for r in gold[:3]:
        cut.append(_bgzf(bamio.bgzf_decompress(r), 6, zlib.Z_DEFAULT_STRATEGY, block=0x1234) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    fields = ("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig")

    def decode(files):
        ctx.set_profiling(True)
        s, fo = ctx.bam_decode(files)
        kt = ctx.kernel_times()
        got = ctx.soa_to_numpy(s, fields=fields)
        idx = np.arange(0, s.n_records, max(1, s.n_records // 50), dtype=np.uint32)
        blob, off = ctx.bam_records(idx)
        ctx.bam_release()
        ctx.set_profiling(False)
        return fo.copy(), got, bytes(blob), kt

    fo_m, got_m, blob_m, kt_m = decode(gold)
    assert "bam_index" in kt_m and "bam_index_chain" not in kt_m         # htslib's blocks: no chain walked
    tbk_debug(monkeypatch, index_chain=1)
    fo_c, got_c, blob_c, kt_c = decode(gold)
    tbk_debug(monkeypatch, index_chain=None)
    assert "bam_index_chain" in kt_c and np.array_equal(fo_m, fo_c) and blob_m == blob_c
    for name in fields:
        assert np.array_equal(got_m[name], got_c[name]), name
    fo_x, got_x, blob_x, kt_x = decode(cut + gold[3:])                     # three files need the chain: every file takes it
    assert "bam_index_chain" in kt_x and np.array_equal(fo_x, fo_m) and blob_x == blob_m
    for name in fields:
        assert np.array_equal(got_x[name], got_m[name]), name


def test_tile_join_of_a_device_and_a_host_part(ctx, bam_loader):
    """tbk_tile_join: the first files decoded on the GPU (tbk_bam_decode), the rest by the host — one device tile that equals the tile
    of all the files array for array, collapses to the oracle's groups, and whose device-side records are the ones tbk_bam_records
    hands back; parts with carried tags are refused"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, soa
    paths = sample_paths("t2")
    bams = [bam_loader(p) for p in paths]
    whole = soa.tile_from_bams(bams)
    kd = 4
    s, fo_d = ctx.bam_decode([open(p, "rb").read() for p in paths[:kd]])
    ctx.reserve_tile(whole.n_records, whole.cig.shape[0])
    joined, fo = ctx.tile_join(s, soa.tile_from_bams(bams[kd:]))
    assert np.array_equal(fo, whole.file_off) and joined.n_records == whole.n_records and joined.n_cigar_ops == whole.cig.shape[0]
    got = ctx.soa_to_numpy(joined)
    for name in ("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig"):
        assert np.array_equal(got[name], getattr(whole, name)), name
    g = api.to_numpy(ctx.collapse_struct(joined, len(paths)))
    want = orc.collapse(whole)
    assert g["n_groups"] == want["n_groups"] and g["n_passed"] == want["n_passed"]
    for k_ in ("rep", "yc", "yx", "yd", "g_start", "g_end"):
        assert np.array_equal(np.asarray(g[k_]).astype(np.float64), np.asarray(want[k_]).astype(np.float64)), k_
    n_d = int(fo_d[-1])
    idx = np.array([0, n_d - 1, n_d // 2], np.uint32)
    blob, off = ctx.bam_records(idx)
    assert len(blob) == int(off[-1]) > 0
    merged = soa.tile_from_bams([bam_loader(os.path.join(GOLDEN, "t1", "t1.bam"))])      # a TieBrush-merged part carries YC / YX / YD
    with pytest.raises(api.TbkError):
        ctx.tile_join(s, merged)
    ctx.bam_release()


def test_bam_decode_then_collapse_equals_oracle(ctx, bam_loader):
    """compressed bytes -> device tile -> tbk_collapse_tile, nothing decoded on the host: the golden t2 collapse"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, soa
    paths = sample_paths("t2")
    raw = [open(p, "rb").read() for p in paths]
    tile = soa.tile_from_bams([bam_loader(p) for p in paths])
    want = orc.collapse(tile)
    s, fo = ctx.bam_decode(raw)
    got = api.to_numpy(ctx.collapse_struct(s, len(paths)))
    assert got["n_groups"] == want["n_groups"] and got["n_passed"] == want["n_passed"]
    for k in ("rep", "yc", "yx", "yd", "g_start", "g_end"):
        assert np.array_equal(np.asarray(got[k]).astype(np.float64), np.asarray(want[k]).astype(np.float64)), k
    ctx.bam_release()


def test_bam_decode_refuses_malformed_records(ctx, tmp_path):
    import struct
    from tiebrush_amd import api, bamio
    body = b"".join(bamio.encode_record(0, 100 + i, 0, 60, [(50 << 4)], b"r%d" % i, b"NHC\x01") for i in range(50))
    raw = bytearray(bamio.build_bam("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:1000\n", ["c"], [1000], body))
    _, p = bamio.parse_header(bytes(raw))
    good = bamio.bgzf_compress(bytes(raw), 1) + bamio._BGZF_EOF
    s, fo = ctx.bam_decode([good])
    assert s.n_records == 50
    for what in ("l_read_name", "tid", "truncated"):
        b = bytearray(raw)
        if what == "l_read_name":
            b[p + 4 + 8] = 250
        elif what == "tid":
            b[p + 4:p + 8] = struct.pack("<i", 9)
        else:
            del b[-7:]
        with pytest.raises(api.TbkError):
            ctx.bam_decode([bamio.bgzf_compress(bytes(b), 1) + bamio._BGZF_EOF])
    ctx.bam_release()
