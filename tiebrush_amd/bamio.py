"""Minimal pure-Python/numpy BGZF + BAM codec (zlib + struct only).

Tooling for tests, fixture pinning and the synthetic-input generator.  It is an
independent third implementation next to the C++ host codec
(`tiebrush_amd/csrc/host/bam.cpp`) and is NOT on the product hot path.

Format reference: SAM/BAM specification (BGZF = concatenated gzip members with a
`BC` extra sub-field; BAM record layout as listed in SURVEY.md A.1).

What it extracts is exactly the structure-of-arrays view that the C-ABI
(`include/tbk.h`) consumes, with the reference's aux-tag defaults applied where
the reference applies them on read:
  * strand  : GSamRecord::spliceStrand      (/root/reference/src/GSam.cpp:464-475)
  * nh      : tag_int("NH") / tag_int("NH",1) (tiebrush.cpp:537, :397); absent -> NH_ABSENT
  * yc/yx/yd: tag_float("YC"), tag_int("YX",1), tag_int("YD",0) (tiebrush.cpp:389-395)
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

NH_ABSENT = -(2**31)
CIGAR_OPS = "MIDNSHP=XB"

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


# --------------------------------------------------------------------------- BGZF
def bgzf_decompress(data: bytes) -> bytes:
    """Inflate every BGZF member of `data` and concatenate the payloads."""
    out = []
    off = 0
    n = len(data)
    mv = memoryview(data)
    while off < n:
        if data[off:off + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF member at offset %d" % off)
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        # find BC subfield
        p = off + 12
        end = p + xlen
        bsize = None
        while p < end:
            si1, si2, slen = data[p], data[p + 1], struct.unpack_from("<H", data, p + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", data, p + 4)[0]
            p += 4 + slen
        if bsize is None:
            raise ValueError("BGZF member without BC subfield")
        cdata = mv[off + 12 + xlen: off + bsize + 1 - 8]
        isize = struct.unpack_from("<I", data, off + bsize + 1 - 4)[0]
        if isize:
            out.append(zlib.decompress(cdata, -15, isize))
        off += bsize + 1
    return b"".join(out)


def bgzf_compress(payload: bytes, level: int = 6) -> bytes:
    """Deflate `payload` into <=0xff00-byte BGZF members and append the EOF block."""
    out = []
    for i in range(0, len(payload), 0xFF00):
        chunk = payload[i:i + 0xFF00]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        bsize = len(c) + 25
        out.append(struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, bsize))
        out.append(c)
        out.append(struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    out.append(_BGZF_EOF)
    return b"".join(out)


# --------------------------------------------------------------------------- BAM
@dataclass
class BamHeader:
    text: str
    ref_names: list
    ref_lens: list

    def co_samples(self):
        """@CO SAMPLE:<x> lines (commons.h:23-71)."""
        res = []
        for line in self.text.split("\n"):
            if line.startswith("@CO\tSAMPLE:"):
                res.append(line[len("@CO\tSAMPLE:"):])
        return res

    def is_tiebrush(self):
        """@PG with PN:TieBrush and a VN tag (tmerge.cpp:70-77)."""
        for line in self.text.split("\n"):
            if line.startswith("@PG"):
                f = line.split("\t")[1:]
                if "PN:TieBrush" in f and any(x.startswith("VN:") for x in f):
                    return True
        return False


@dataclass
class BamSoA:
    """Structure-of-arrays view of one BAM file (record order = file order)."""
    header: BamHeader
    n: int
    tid: np.ndarray
    pos: np.ndarray
    flag: np.ndarray
    mapq: np.ndarray
    cig_off: np.ndarray
    cig: np.ndarray
    strand: np.ndarray        # ASCII '+','-','.'
    nh: np.ndarray            # NH_ABSENT when the tag is missing
    has_yc: np.ndarray
    yc: np.ndarray            # float64, 0.0 when absent (bam_aux2f semantics)
    yx: np.ndarray            # int64, 1 when absent
    yd: np.ndarray            # int64, 0 when absent
    qname: list = field(default_factory=list)
    md: list = field(default_factory=list)       # bytes or None
    rec_off: np.ndarray = None                   # offset of each record's block_size field
    raw: bytes = b""                             # inflated file
    aux_types: list = field(default_factory=list)  # per record dict tag->type char (only when keep_aux)


_AUX_FIXED = {ord("A"): 1, ord("c"): 1, ord("C"): 1, ord("s"): 2, ord("S"): 2,
              ord("i"): 4, ord("I"): 4, ord("f"): 4, ord("d"): 8}
_AUX_FMT = {ord("c"): "<b", ord("C"): "<B", ord("s"): "<h", ord("S"): "<H", ord("i"): "<i",
            ord("I"): "<I", ord("f"): "<f", ord("d"): "<d"}
_B_SIZE = {ord("c"): 1, ord("C"): 1, ord("s"): 2, ord("S"): 2, ord("i"): 4, ord("I"): 4, ord("f"): 4}


def iter_aux(raw, p, end):
    """Yield (tag_bytes, type_byte, value_offset, next_offset)."""
    while p + 3 <= end:
        tag = raw[p:p + 2]
        t = raw[p + 2]
        v = p + 3
        if t in _AUX_FIXED:
            nx = v + _AUX_FIXED[t]
        elif t == 90 or t == 72:  # Z / H
            nx = raw.index(b"\0", v) + 1
        elif t == 66:  # B
            st = raw[v]
            cnt = struct.unpack_from("<I", raw, v + 1)[0]
            nx = v + 5 + cnt * _B_SIZE[st]
        else:
            raise ValueError("bad aux type %r" % chr(t))
        yield tag, t, v, nx
        p = nx


def aux_to_int(raw, t, v):
    """htslib bam_aux2i: integer types only, anything else -> 0."""
    if t in (99, 67, 115, 83, 105, 73):
        return struct.unpack_from(_AUX_FMT[t], raw, v)[0]
    return 0


def aux_to_float(raw, t, v):
    """htslib bam_aux2f: d, f, or integer types; anything else -> 0."""
    if t == 100 or t == 102:
        return float(struct.unpack_from(_AUX_FMT[t], raw, v)[0])
    return float(aux_to_int(raw, t, v))


def parse_header(raw: bytes):
    if raw[:4] != b"BAM\1":
        raise ValueError("not a BAM stream")
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].split(b"\0")[0].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    names, lens = [], []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        names.append(raw[p + 4:p + 4 + l_name - 1].decode())
        lens.append(struct.unpack_from("<i", raw, p + 4 + l_name)[0])
        p += 8 + l_name
    return BamHeader(text, names, lens), p


def read_bam(path: str, keep_names: bool = True, keep_md: bool = False, keep_aux: bool = False) -> BamSoA:
    with open(path, "rb") as fh:
        raw = bgzf_decompress(fh.read())
    return parse_bam(raw, keep_names=keep_names, keep_md=keep_md, keep_aux=keep_aux)


def parse_bam(raw: bytes, keep_names=True, keep_md=False, keep_aux=False) -> BamSoA:
    hdr, p = parse_header(raw)
    n_raw = len(raw)
    offs = []
    unpack_i = struct.Struct("<i").unpack_from
    while p < n_raw:
        offs.append(p)
        p += 4 + unpack_i(raw, p)[0]
    n = len(offs)
    off = np.asarray(offs, dtype=np.int64)
    buf = np.frombuffer(raw, dtype=np.uint8)

    def gather(o, dt):
        w = np.dtype(dt).itemsize
        idx = (off + o)[:, None] + np.arange(w)[None, :]
        return buf[idx].copy().view(dt).reshape(-1)

    if n == 0:
        z = np.zeros(0, np.int32)
        return BamSoA(hdr, 0, z, z, z.astype(np.uint16), z.astype(np.uint8), np.zeros(1, np.uint32),
                      z.astype(np.uint32), z.astype(np.uint8), z, z.astype(bool), z.astype(np.float64),
                      z.astype(np.int64), z.astype(np.int64), [], [], off, raw)
    block = gather(0, "<i4")
    tid = gather(4, "<i4")
    pos = gather(8, "<i4")
    l_name = gather(12, "u1").astype(np.int64)
    mapq = gather(13, "u1")
    n_cig = gather(16, "<u2").astype(np.int64)
    flag = gather(18, "<u2")
    l_seq = gather(20, "<i4").astype(np.int64)
    cig_start = off + 36 + l_name
    cig_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(n_cig, out=cig_off[1:])
    tot = int(cig_off[-1])
    if tot:
        rec_of = np.repeat(np.arange(n), n_cig)
        within = np.arange(tot) - cig_off[rec_of]
        bpos = cig_start[rec_of] + 4 * within
        idx = bpos[:, None] + np.arange(4)[None, :]
        cig = buf[idx].copy().view("<u4").reshape(-1)
    else:
        cig = np.zeros(0, np.uint32)
    aux_start = cig_start + 4 * n_cig + (l_seq + 1) // 2 + l_seq
    rec_end = off + 4 + block

    strand = np.full(n, ord("."), dtype=np.uint8)
    nh = np.full(n, NH_ABSENT, dtype=np.int64)
    has_yc = np.zeros(n, dtype=bool)
    yc = np.zeros(n, dtype=np.float64)
    yx = np.ones(n, dtype=np.int64)
    yd = np.zeros(n, dtype=np.int64)
    qnames = []
    mds = []
    auxt = []
    a0 = aux_start.tolist()
    a1 = rec_end.tolist()
    fl = flag.tolist()
    o_l = offs
    ln = l_name.tolist()
    for i in range(n):
        if keep_names:
            qnames.append(raw[o_l[i] + 36:o_l[i] + 36 + ln[i] - 1])
        xs = 0
        ts = 0
        md = None
        types = {} if keep_aux else None
        seen = 0
        for tag, t, v, _nx in iter_aux(raw, a0[i], a1[i]):
            if keep_aux:
                types.setdefault(tag, chr(t))
            # htslib bam_aux_get returns the FIRST occurrence of a tag
            if tag == b"NH" and not (seen & 1):
                seen |= 1
                nh[i] = aux_to_int(raw, t, v)
            elif tag == b"XS" and not (seen & 2):
                seen |= 2
                xs = raw[v] if t in (65, 90) else 0   # tag_char1: A or Z -> first char (GSam.cpp:436-444)
            elif tag == b"ts" and not (seen & 4):
                seen |= 4
                ts = raw[v] if t in (65, 90) else 0
            elif tag == b"YC" and not (seen & 8):
                seen |= 8
                has_yc[i] = True
                yc[i] = aux_to_float(raw, t, v)
            elif tag == b"YX" and not (seen & 16):
                seen |= 16
                yx[i] = aux_to_int(raw, t, v)
            elif tag == b"YD" and not (seen & 32):
                seen |= 32
                yd[i] = aux_to_int(raw, t, v)
            elif tag == b"MD" and keep_md and not (seen & 64):
                seen |= 64
                if t == 90:
                    md = raw[v:raw.index(b"\0", v)]
        c = xs
        if c == 0 and ts in (43, 45):  # spliceStrand, GSam.cpp:464-475
            c = (45 if ts == 43 else 43) if (fl[i] & 0x10) else ts
        strand[i] = c if c in (43, 45) else 46
        if keep_md:
            mds.append(md)
        if keep_aux:
            auxt.append(types)
    return BamSoA(hdr, n, tid.astype(np.int32), pos.astype(np.int32), flag.astype(np.uint16),
                  mapq.astype(np.uint8), cig_off.astype(np.uint32), cig.astype(np.uint32), strand,
                  nh, has_yc, yc, yx, yd, qnames, mds, off, raw, auxt)


def record_identity(soa: BamSoA, i: int) -> bytes:
    """Bytes of record i from refID up to (not including) aux: what `samtools view`
    shows apart from tags (the reference's run_tests.sh:14 comparison minus tags)."""
    o = int(soa.rec_off[i])
    raw = soa.raw
    l_name = raw[o + 12]
    n_cig = struct.unpack_from("<H", raw, o + 16)[0]
    l_seq = struct.unpack_from("<i", raw, o + 20)[0]
    aux = o + 36 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
    # skip bin (o+14..16): recomputed by some writers; everything else verbatim
    return raw[o + 4:o + 14] + raw[o + 16:aux]


def record_bytes(soa: BamSoA, i: int) -> bytes:
    """The whole record i (refID .. end of aux, without its block_size field)."""
    o = int(soa.rec_off[i])
    bs = struct.unpack_from("<I", soa.raw, o)[0]
    return bytes(soa.raw[o + 4:o + 4 + bs])


def record_aux(soa: BamSoA, i: int):
    """Ordered list of (tag, type_char, python_value) of record i."""
    o = int(soa.rec_off[i])
    raw = soa.raw
    block = struct.unpack_from("<i", raw, o)[0]
    l_name = raw[o + 12]
    n_cig = struct.unpack_from("<H", raw, o + 16)[0]
    l_seq = struct.unpack_from("<i", raw, o + 20)[0]
    aux = o + 36 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
    res = []
    for tag, t, v, nx in iter_aux(raw, aux, o + 4 + block):
        if t in _AUX_FMT:
            val = struct.unpack_from(_AUX_FMT[t], raw, v)[0]
        elif t == 65:
            val = chr(raw[v])
        elif t in (90, 72):
            val = raw[v:nx - 1].decode()
        else:
            val = raw[v:nx]
        res.append((tag.decode(), chr(t), val))
    return res


# --------------------------------------------------------------------------- writer
def build_bam(header_text: str, ref_names, ref_lens, records: bytes) -> bytes:
    """Assemble an uncompressed BAM stream from header parts + pre-encoded records."""
    ht = header_text.encode()
    parts = [b"BAM\1", struct.pack("<i", len(ht)), ht, struct.pack("<i", len(ref_names))]
    for nm, ln in zip(ref_names, ref_lens):
        b = nm.encode() + b"\0"
        parts.append(struct.pack("<i", len(b)) + b + struct.pack("<i", ln))
    parts.append(records)
    return b"".join(parts)


def reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def encode_record(tid, pos, flag, mapq, cigar, qname: bytes, aux: bytes = b"", l_seq: int = 0,
                  seq: bytes = b"", qual: bytes = b"", mtid=-1, mpos=-1, tlen=0, ref_len=None) -> bytes:
    """Encode one BAM record.  `cigar` is a sequence of uint32 (len<<4|op)."""
    nm = qname + b"\0"
    if ref_len is None:
        ref_len = sum((c >> 4) for c in cigar if (c & 0xF) in (0, 2, 3, 7, 8))
    b = reg2bin(pos, pos + max(ref_len, 1)) if pos >= 0 else 4680
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(nm), mapq, b, len(cigar), flag, l_seq, mtid, mpos, tlen)
    body += nm + struct.pack("<%dI" % len(cigar), *cigar) + seq + qual + aux
    return struct.pack("<i", len(body)) + body


def write_bam(path: str, header_text: str, ref_names, ref_lens, records: bytes, level: int = 1):
    with open(path, "wb") as fh:
        fh.write(bgzf_compress(build_bam(header_text, ref_names, ref_lens, records), level))
