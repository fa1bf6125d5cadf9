"""BAM -> SAM text for the tests (an independent formatter over the raw BAM bytes; the product's SAM reader is
tiebrush_amd/csrc/host/sam.cpp)."""
import struct

from tiebrush_amd import bamio

_SEQ = "=ACMGRSVTWYHKDBN"


def _aux_text(raw, p, end):
    out = []
    while p < end:
        tag = raw[p:p + 2].decode()
        t = chr(raw[p + 2])
        p += 3
        if t == "A":
            out.append("%s:A:%s" % (tag, chr(raw[p]))); p += 1
        elif t in "cCsSiI":
            fmt, sz = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4), "I": ("<I", 4)}[t]
            out.append("%s:i:%d" % (tag, struct.unpack_from(fmt, raw, p)[0])); p += sz
        elif t == "f":
            out.append("%s:f:%s" % (tag, repr(struct.unpack_from("<f", raw, p)[0]))); p += 4
        elif t in "ZH":
            e = raw.index(b"\0", p)
            out.append("%s:%s:%s" % (tag, t, raw[p:e].decode())); p = e + 1
        elif t == "B":
            st = chr(raw[p]); n = struct.unpack_from("<I", raw, p + 1)[0]; p += 5
            fmt, sz = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4), "I": ("<I", 4), "f": ("<f", 4)}[st]
            vals = [struct.unpack_from(fmt, raw, p + i * sz)[0] for i in range(n)]
            out.append("%s:B:%s%s" % (tag, st, "".join(",%s" % (repr(v) if st == "f" else v) for v in vals))); p += n * sz
        else:
            raise ValueError("aux type %r" % t)
    return out


def bam_to_sam_text(path):
    """-> (sam text, list of raw record byte strings without block_size)"""
    raw = bamio.bgzf_decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode().rstrip("\0")
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    names = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, p)[0]
        names.append(raw[p + 4:p + 4 + ln - 1].decode()); p += 4 + ln + 4
    lines = [text if text.endswith("\n") or not text else text + "\n"]
    recs = []
    while p < len(raw):
        bs = struct.unpack_from("<i", raw, p)[0]
        r = raw[p + 4:p + 4 + bs]; recs.append(r); p += 4 + bs
        tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", r, 0)
        q = 32
        qname = r[q:q + l_rn - 1].decode(); q += l_rn
        cig = "".join("%d%s" % (w >> 4, bamio.CIGAR_OPS[w & 15]) for w in struct.unpack_from("<%dI" % n_cig, r, q)) or "*"; q += 4 * n_cig
        seq = "".join(_SEQ[(r[q + (i >> 1)] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq)) or "*"; q += (l_seq + 1) // 2
        ql = r[q:q + l_seq]; q += l_seq
        qual = "*" if (l_seq == 0 or all(b == 0xFF for b in ql)) else "".join(chr(b + 33) for b in ql)
        rname = names[tid] if tid >= 0 else "*"
        rnext = "*" if mtid < 0 else ("=" if mtid == tid else names[mtid])
        f = [qname, str(flag), rname, str(pos + 1), str(mapq), cig, rnext, str(mpos + 1), str(tlen), seq, qual] + _aux_text(r, q, len(r))
        lines.append("\t".join(f) + "\n")
    return "".join(lines), recs
