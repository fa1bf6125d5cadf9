"""CPU tests of the host-side C++ (BGZF/BAM codec, TInputFiles mirror, SoA tile loader, htslib tag rules)
against the independent Python codec and the oracle.  No GPU involved."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, sample_paths

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.environ.get("TBK_TEST_TBH_TOOL") or os.path.join(ROOT, "tiebrush_amd", "_build", "tbh_tool")   # (the override: the sanitizer builds of tools/san_check.sh)


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tiebrush_amd", "csrc", "host"), "-s",
                        os.path.join("..", "..", "_build", "tbh_tool")], check=True)


def test_bam_roundtrip_is_byte_identical_after_inflate(tmp_path):
    from tiebrush_amd import bamio
    for src in (os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2s2.bam")):
        out = str(tmp_path / "o.bam")
        subprocess.run([TOOL, "cat", src, out], check=True)
        a = bamio.bgzf_decompress(open(src, "rb").read())
        b = bamio.bgzf_decompress(open(out, "rb").read())
        ha, pa = bamio.parse_header(a)
        hb, pb = bamio.parse_header(b)
        assert ha.text.rstrip("\n") == hb.text.rstrip("\n") and ha.ref_names == hb.ref_names and ha.ref_lens == hb.ref_lens
        assert a[pa:] == b[pb:]
        assert open(out, "rb").read().endswith(bamio._BGZF_EOF)


def test_merge_order_matches_oracle(bam_loader):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import soa
    paths = sample_paths("t2")[:5]
    bams = [bam_loader(p) for p in paths]
    tile = soa.tile_from_bams(bams)
    want = orc.collapse(tile, want_merge_order=True)["merge_order"]
    fo = tile.file_of()
    want_pairs = np.stack([fo[want], want - tile.file_off[fo[want]]], axis=1)
    out = subprocess.run([TOOL, "mergeorder"] + paths, check=True, capture_output=True, text=True).stdout
    got = np.array([[int(x) for x in l.split()] for l in out.splitlines()], dtype=np.int64)
    assert np.array_equal(got, want_pairs)


def test_soa_tile_matches_python_decoder(tmp_path, bam_loader):
    from tiebrush_amd import soa
    paths = [os.path.join(GOLDEN, "t1", "t1.bam")] + sample_paths("t2")[:2]
    bams = [bam_loader(p, keep_md=True) for p in paths]
    tile = soa.tile_from_bams(bams, with_names=True, with_md=True)
    d = str(tmp_path)
    subprocess.run([TOOL, "soa", d] + paths, check=True)

    def rd(name, dt):
        return np.fromfile(os.path.join(d, name), dtype=dt)

    assert np.array_equal(rd("file_off", np.uint32), tile.file_off)
    assert rd("tbmerged", np.uint8).tolist() == [1, 0, 0]
    for name, dt in (("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("strand", np.uint8),
                     ("nh", np.int32), ("cig_off", np.uint32), ("cig", np.uint32), ("md_off", np.uint32), ("md", np.uint8),
                     ("md_has", np.uint8), ("qname_hash", np.uint64)):
        assert np.array_equal(rd(name, dt), getattr(tile, name)), name
    assert np.array_equal(rd("qname_off", np.uint32), tile.qn_off) and np.array_equal(rd("qname", np.uint8), tile.qn)   # -A compares the names
    # carried tags are only meaningful for TieBrush-merged files
    n0 = int(tile.file_off[1])
    assert np.array_equal(rd("yc_in", np.float64)[:n0], tile.yc_in[:n0])
    assert np.array_equal(rd("yx_in", np.int64)[:n0], tile.yx_in[:n0])
    assert np.array_equal(rd("yd_in", np.int64)[:n0], tile.yd_in[:n0])
    hdr = open(os.path.join(d, "header.txt")).read()
    co = [l for l in hdr.split("\n") if l.startswith("@CO\tSAMPLE:")]
    assert len(co) == 12 and co[-1].endswith("t2s1.bam") and co[0].endswith("t1s0.bam")
    pg = [l for l in hdr.split("\n") if l.startswith("@PG")]
    assert pg[-1].startswith("@PG\tID:TieBrush.1\tPN:TieBrush\tPP:TieBrush\tVN:test")


def test_whole_input_loader_matches_python_decoder(tmp_path, bam_loader):
    """fastload.cpp (the whole-input path of the tiebrush command line: every member inflated as its own task, records indexed,
    SoA filled in parallel) against the Python decoder: the tile arrays, the carried tags of TieBrush-merged inputs, the raw
    records behind tile indices; unplaced reads at the end of an input are left out"""
    import struct
    from tiebrush_amd import bamio, soa, synth
    from helpers import paired_end_like_files, tile_from_records
    extra = synth.write_bams(tile_from_records(paired_end_like_files()), str(tmp_path / "pe"))   # unmapped mates, unplaced tail, empty input
    paths = [os.path.join(GOLDEN, "t1", "t1.bam")] + sample_paths("t2")[:2] + extra
    bams = [bam_loader(p) for p in paths]
    d = str(tmp_path / "fs")
    os.makedirs(d)
    subprocess.run([TOOL, "fastsoa", d] + paths, check=True)

    def rd(name, dt):
        return np.fromfile(os.path.join(d, name), dtype=dt)

    keep = [np.nonzero(b.tid >= 0)[0] for b in bams]            # (a sorted BAM holds its refID -1 reads at the end)
    assert all(len(k_) == 0 or k_[-1] == len(k_) - 1 for k_ in keep)
    fo = np.concatenate([[0], np.cumsum([len(k_) for k_ in keep])]).astype(np.uint32)
    assert np.array_equal(rd("file_off", np.uint32), fo)
    assert rd("tbmerged", np.uint8).tolist() == [1, 0, 0] + [0] * len(extra)
    for name, dt in (("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("strand", np.uint8), ("nh", np.int32)):
        assert np.array_equal(rd(name, dt), np.concatenate([getattr(b, name)[k_] for b, k_ in zip(bams, keep)])), name
    ncig = np.concatenate([np.diff(b.cig_off.astype(np.int64))[k_] for b, k_ in zip(bams, keep)])
    assert np.array_equal(rd("cig_off", np.uint32), np.concatenate([[0], np.cumsum(ncig)]).astype(np.uint32))
    assert np.array_equal(rd("cig", np.uint32), np.concatenate([b.cig[:int(b.cig_off[len(k_)])] for b, k_ in zip(bams, keep)]))
    n0 = int(fo[1])
    assert np.array_equal(rd("yc_in", np.float64)[:n0], bams[0].yc[:n0].astype(np.float64))
    assert np.array_equal(rd("yx_in", np.int64)[:n0], bams[0].yx[:n0]) and np.array_equal(rd("yd_in", np.int64)[:n0], bams[0].yd[:n0])
    raw = open(os.path.join(d, "probe_records"), "rb").read()
    p, seen = 0, 0
    while p < len(raw):
        g, ln = struct.unpack_from("<II", raw, p)
        f = int(np.searchsorted(fo, g, side="right")) - 1
        assert raw[p + 8:p + 8 + ln] == bamio.record_bytes(bams[f], g - int(fo[f]))
        p += 8 + ln
        seen += 1
    assert seen >= 3 * 3


@pytest.mark.parametrize("val,typ", [(0, "C"), (254, "C"), (255, "S"), (65534, "S"), (65535, "I"), (-1, "c"), (-129, "s"),
                                     (-40000, "i"), (70000, "I")])
def test_int_tag_width_rules_on_append(tmp_path, val, typ):
    """htslib bam_aux_update_int: strict '<' against UINT8_MAX / UINT16_MAX (goldens: YD 255 -> S, 254 -> C)"""
    from tiebrush_amd import bamio
    src = os.path.join(GOLDEN, "t2", "t2s2.bam")
    out = str(tmp_path / "o.bam")
    subprocess.run([TOOL, "tags", src, out, "ZZ=i:%d" % val], check=True)
    b = bamio.read_bam(out)
    aux = bamio.record_aux(b, 0)
    assert aux[-1] == ("ZZ", typ, val)


def test_tag_update_in_place_rules(tmp_path):
    from tiebrush_amd import bamio
    src = os.path.join(GOLDEN, "t1", "t1.bam")   # records carry YC:C (0.0.6 format), some YD:C
    out = str(tmp_path / "o.bam")
    # float update of an integer-typed YC fails (EINVAL) and the stale value survives; YX appended;
    # YD widened in place when needed, deleted with "del"
    subprocess.run([TOOL, "tags", src, out, "YC=f:7.5", "YX=i:3", "YD=i:300"], check=True)
    a, b = bamio.read_bam(src), bamio.read_bam(out)
    for i in (0, 1, 2, 50):
        aa, bb = dict((t, (ty, v)) for t, ty, v in bamio.record_aux(a, i)), dict((t, (ty, v)) for t, ty, v in bamio.record_aux(b, i))
        if "YC" in aa:
            assert bb["YC"] == aa["YC"]
        else:
            assert bb["YC"] == ("f", 7.5)
        assert bb["YX"][1] == 3 and bb["YD"] == ("S", 300)
        order_a = [t for t, _, _ in bamio.record_aux(a, i)]
        order_b = [t for t, _, _ in bamio.record_aux(b, i)]
        assert order_b[:len(order_a)] == order_a     # existing tags keep their positions
    out2 = str(tmp_path / "o2.bam")
    subprocess.run([TOOL, "tags", out, out2, "YD=i:5", "YX=del"], check=True)
    c = bamio.read_bam(out2)
    cc = dict((t, (ty, v)) for t, ty, v in bamio.record_aux(c, 1))
    assert cc["YD"] == ("S", 5) and "YX" not in cc   # old width kept when it is wide enough


def _mini_bam(path, recs, level=1):
    """recs: list of (pos, flag, aux bytes)"""
    from tiebrush_amd import bamio
    body = b"".join(bamio.encode_record(0, pos, flag, 60, [(50 << 4) | 0, (200 << 4) | 3, (50 << 4) | 0], b"r%d" % i, aux)
                    for i, (pos, flag, aux) in enumerate(recs))
    bamio.write_bam(path, "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:100000\n", ["chr1"], [100000], body, level)


def test_splice_strand_branches_through_the_host_codec(tmp_path):
    """GSamRecord::spliceStrand (GSam.cpp:464-475): XS of type A or Z wins (its first byte, even '.', which then blocks the
    ts fallback); else ts:A:+/- counts, flipped on the reverse strand; anything else is '.'.  The C++ loader, the Python
    decoder and the oracle's expectation must agree record by record."""
    from tiebrush_amd import bamio
    cases = [
        (100, 0, b"XSA+", "+"), (110, 16, b"XSA-", "-"), (120, 0, b"XSZ-\0", "-"), (130, 0, b"XSZ+extra\0", "+"),
        (140, 0, b"XSA.", "."), (150, 0, b"XSA." + b"tsA+", "."),          # XS:A:. blocks ts
        (160, 0, b"tsA+", "+"), (170, 16, b"tsA+", "-"), (180, 16, b"tsA-", "+"), (190, 0, b"tsA-", "-"),
        (200, 0, b"tsA?", "."), (210, 0, b"", "."), (220, 0, b"XSi\x05\0\0\0" + b"tsA+", "+"),   # XS of another type is not a strand
        (230, 0, b"NHC\x01" + b"tsA-" + b"XSA+", "+"),                     # XS wins wherever it stands
    ]
    p = str(tmp_path / "s.bam")
    _mini_bam(p, [(pos, fl, aux) for pos, fl, aux, _ in cases])
    want = np.array([ord(c) for _, _, _, c in cases], np.uint8)
    assert np.array_equal(bamio.read_bam(p).strand, want)                  # Python decoder
    d = str(tmp_path / "soa")
    os.makedirs(d)
    subprocess.run([TOOL, "soa", d, p], check=True)                        # C++ loader (GSamRecord::spliceStrand mirror)
    assert np.array_equal(np.fromfile(os.path.join(d, "strand"), np.uint8), want)


def _corrupt(src_bytes, mutate):
    """inflate a BAM, mutate the record stream in place, deflate again"""
    from tiebrush_amd import bamio
    raw = bytearray(bamio.bgzf_decompress(src_bytes))
    _, p = bamio.parse_header(bytes(raw))
    mutate(raw, p)
    return bamio.bgzf_compress(bytes(raw), 1) + bamio._BGZF_EOF


@pytest.mark.parametrize("what", ["l_read_name", "n_cigar", "l_seq_negative", "tid", "no_nul", "truncated", "crc", "bsize"])
def test_malformed_input_is_refused_not_read_past(tmp_path, what):
    """what htslib rejects, the host codec rejects: field lengths beyond the record, reference ids outside the header,
    a read name without its NUL, a truncated record stream, a BGZF member with a wrong CRC32 or an impossible BSIZE"""
    import struct
    good = str(tmp_path / "g.bam")
    _mini_bam(good, [(100 + 10 * i, 0, b"NHC\x01") for i in range(20)])
    data = open(good, "rb").read()

    def mut(raw, p):
        r = p + 4 + 0  # first record: refID at r, l_read_name at r+8, n_cigar_op at r+12, l_seq at r+16
        if what == "l_read_name":
            raw[r + 8] = 250
        elif what == "n_cigar":
            raw[r + 12:r + 14] = struct.pack("<H", 60000)
        elif what == "l_seq_negative":
            raw[r + 16:r + 20] = struct.pack("<i", -5)
        elif what == "tid":
            raw[r:r + 4] = struct.pack("<i", 7)
        elif what == "no_nul":
            raw[r + 32 + raw[r + 8] - 1] = ord("x")
        elif what == "truncated":
            del raw[-9:]

    if what == "crc":
        b = bytearray(data)
        bs = struct.unpack("<H", b[16:18])[0]
        b[bs + 1 - 8] ^= 0xFF                                   # CRC32 of the first member
        bad = bytes(b)
    elif what == "bsize":
        b = bytearray(data)
        b[16:18] = struct.pack("<H", 10)                        # BSIZE smaller than header + trailer
        bad = bytes(b)
    else:
        bad = _corrupt(data, mut)
    p = str(tmp_path / "bad.bam")
    open(p, "wb").write(bad)
    d = str(tmp_path / "soa")
    os.makedirs(d)
    for cmd in ("soa", "fastsoa"):                                # the streaming reader and the whole-input loader
        r = subprocess.run([TOOL, cmd, d, p], capture_output=True, text=True)
        assert r.returncode != 0 and r.returncode > 0, (what, cmd, r.returncode, r.stderr)   # a clean error exit, not a signal
        assert r.stderr.strip() != ""
        ok = subprocess.run([TOOL, cmd, d, good], capture_output=True, text=True)
        assert ok.returncode == 0


def test_streamed_tiles_cut_where_no_read_crosses(tmp_path, bam_loader):
    """TInputFiles::next_tile: the tiles partition every input in order; at every cut no read of ANY input that starts
    before it reaches it (a global bundle boundary or a reference change), so tiles collapse independently; and the
    windows stay near the requested tile size instead of holding the files."""
    from tiebrush_amd import synth
    tile = synth.make_tile(6, 4000, "c3", n_loci=400)                 # many separate loci: plenty of places to cut
    paths = synth.write_bams(tile, str(tmp_path / "s"))
    bams = [bam_loader(p) for p in paths]
    out = subprocess.run([TOOL, "tiles", "3000"] + paths, check=True, capture_output=True, text=True).stdout
    rows = np.array([[int(x) for x in l.split()] for l in out.splitlines()], dtype=np.int64)
    assert len(rows) > 2                                              # really streamed in several tiles
    taken = rows[:, 1:]
    placed = [int((b.tid >= 0).sum()) for b in bams]
    assert taken.sum(0).tolist() == placed                            # every placed record, once
    cum = np.cumsum(taken, axis=0)
    for t in range(len(rows) - 1):
        # cut key = the smallest start among the records right behind the cut
        nxt = [((int(b.tid[c]) + 1) << 32) | (int(b.pos[c]) + 1) for b, c in zip(bams, cum[t]) if c < b.n and b.tid[c] >= 0]
        cut = min(nxt)
        for b, c in zip(bams, cum[t]):
            if c == 0:
                continue
            ops, ln = b.cig & 0xF, (b.cig >> 4).astype(np.int64)
            ref = np.where(np.isin(ops, [0, 2, 3, 7, 8]), ln, 0)
            csum = np.concatenate([[0], np.cumsum(ref)])
            co = b.cig_off.astype(np.int64)
            end = b.pos[:c].astype(np.int64) + (csum[co[1:c + 1]] - csum[co[:c]])
            key_end = ((b.tid[:c].astype(np.int64) + 1) << 32) | end
            assert int(key_end.max()) < cut                           # nothing before the cut reaches it
    assert int(rows[-1, 0]) < int(rows[0, 0])                         # consumed records leave the windows
