"""Deterministic synthetic sorted-BAM workloads, produced directly as SoA tiles.

Header: @HD SO:coordinate; contigs chr1 50 Mb, chr2 30 Mb, chr3 20 Mb.  A shared
"transcriptome" of loci (seed 0x71EB) with 1-8 exons (80-400 bp) and introns of 70-5000 bp;
per file (seed 0x71EB0000 + file index) read starts are drawn from a Zipf(1.1)-weighted locus
choice plus a uniform transcript offset; reads are 100 bp.  Profiles:
  c2: CIGAR-only collapse; 100M or aMbNcM; flags {0,16}; NH=1; XS:A on spliced reads only
  c3: c2 + 10 % soft-clipped reads (xS..yS, x,y in [0,8])            -> exercises --clip
  c5: c2 + 3 % 1I/2D inside exons, NH in {1,2,5,20}, MAPQ in {0,1,60},
      1 % 0x100, 0.5 % 0x800                                        -> --exon -N 5 -Q 1
Never emits = X H P ops nor n_cigar > 255 (tiecov would abort / hang on those).
"""
from __future__ import annotations

import numpy as np

from .soa import SoATile, CovInput

REF_NAMES = ["chr1", "chr2", "chr3"]
REF_LENS = [50_000_000, 30_000_000, 20_000_000]
READ_LEN = 100
M, I, D, N, S = 0, 1, 2, 3, 4
MAX_EXONS = 8


def make_transcriptome(n_loci=20000, seed=0x71EB):
    rng = np.random.Generator(np.random.PCG64(seed))
    tid = rng.choice(3, size=n_loci, p=[0.5, 0.3, 0.2]).astype(np.int32)
    n_ex = rng.integers(1, MAX_EXONS + 1, size=n_loci)
    ex_len = rng.integers(80, 401, size=(n_loci, MAX_EXONS))
    in_len = rng.integers(70, 5001, size=(n_loci, MAX_EXONS))
    valid = np.arange(MAX_EXONS)[None, :] < n_ex[:, None]
    ex_len = np.where(valid, ex_len, 0)
    in_len = np.where(np.arange(MAX_EXONS)[None, :] < (n_ex[:, None] - 1), in_len, 0)
    span = ex_len.sum(1) + in_len.sum(1)
    reflen = np.asarray(REF_LENS)[tid]
    start = (rng.random(n_loci) * (reflen - span - 1000)).astype(np.int64) + 500
    # genomic start of each exon and transcript offset of each exon
    g = np.zeros((n_loci, MAX_EXONS), dtype=np.int64)
    t = np.zeros((n_loci, MAX_EXONS + 1), dtype=np.int64)
    g[:, 0] = start
    for e in range(1, MAX_EXONS):
        g[:, e] = g[:, e - 1] + ex_len[:, e - 1] + in_len[:, e - 1]
    t[:, 1:] = np.cumsum(ex_len, axis=1)
    strand = np.where(rng.random(n_loci) < 0.5, ord("+"), ord("-")).astype(np.uint8)
    tlen = t[:, -1]
    # loci shorter than a read get padded (single long exon) so that every locus can emit reads
    short = tlen < READ_LEN + 16
    ex_len[short, 0] += READ_LEN + 16
    t[:, 1:] = np.cumsum(ex_len, axis=1)
    for e in range(1, MAX_EXONS):
        g[:, e] = g[:, e - 1] + ex_len[:, e - 1] + in_len[:, e - 1]
    w = 1.0 / np.arange(1, n_loci + 1) ** 1.1
    w /= w.sum()
    return dict(tid=tid, n_ex=n_ex, ex_len=ex_len, g=g, t=t, strand=strand, tlen=t[:, -1], cdf=np.cumsum(w))


def _reads_for_file(tx, n_reads, fidx, profile, seed_base):
    rng = np.random.Generator(np.random.PCG64(seed_base + fidx))
    loc = np.searchsorted(tx["cdf"], rng.random(n_reads)).clip(0, len(tx["cdf"]) - 1)
    tlen = tx["tlen"][loc]
    clipL = np.zeros(n_reads, dtype=np.int64)
    clipR = np.zeros(n_reads, dtype=np.int64)
    if profile in ("c3",):
        sc = rng.random(n_reads) < 0.10
        clipL = np.where(sc, rng.integers(0, 9, n_reads), 0)
        clipR = np.where(sc, rng.integers(0, 9, n_reads), 0)
    mlen = READ_LEN - clipL - clipR              # aligned (M + I) bases
    off = (rng.random(n_reads) * (tlen - mlen + 1)).astype(np.int64)   # transcript offset of first aligned base
    # exon containing `off`
    t = tx["t"][loc]                              # [n, 9]
    e0 = (off[:, None] >= t[:, 1:]).sum(1)        # index of exon holding the first base
    ex_len = tx["ex_len"][loc]
    g = tx["g"][loc]
    rows = np.arange(n_reads)
    pos = g[rows, e0] + (off - t[rows, e0])       # 0-based genomic position
    # up to 3 M segments
    rem = mlen.copy()
    seg_len = np.zeros((n_reads, 3), dtype=np.int64)
    gap_len = np.zeros((n_reads, 2), dtype=np.int64)
    avail = t[rows, e0 + 1] - off
    e = e0.copy()
    for k in range(3):
        take = np.minimum(rem, avail)
        seg_len[:, k] = take
        rem = rem - take
        more = rem > 0
        if k < 2:
            en = np.minimum(e + 1, MAX_EXONS - 1)
            gap = g[rows, en] - (g[rows, e] + ex_len[rows, e])
            gap_len[:, k] = np.where(more, gap, 0)
            avail = np.where(more, ex_len[rows, en], 0)
            e = np.where(more, en, e)
    # a read that would need a 4th segment is shortened (right clip grows) — rare
    clipR = clipR + rem
    spliced = gap_len[:, 0] > 0
    # indels (c5): 1I or 2D inside the first segment when it is long enough
    ins = np.zeros(n_reads, dtype=bool)
    dele = np.zeros(n_reads, dtype=bool)
    if profile == "c5":
        r = rng.random(n_reads)
        ok = seg_len[:, 0] >= 40
        ins = ok & (r < 0.015)
        dele = ok & (r >= 0.015) & (r < 0.03)
    # ops table [n, 12]: S M (I|D M)? (N M){0..2} S
    ops = np.zeros((n_reads, 12), dtype=np.uint32)
    use = np.zeros((n_reads, 12), dtype=bool)
    c = 0
    ops[:, c] = (clipL << 4) | S
    use[:, c] = clipL > 0
    c += 1
    first = seg_len[:, 0]
    half = first // 2
    m_a = np.where(ins | dele, half, first)
    ops[:, c] = (m_a << 4) | M
    use[:, c] = m_a > 0
    c += 1
    ops[:, c] = np.where(ins, (1 << 4) | I, (2 << 4) | D)
    use[:, c] = ins | dele
    c += 1
    # after 1I the read has consumed one more query base; keep the reference length: M(first-half-1) for ins,
    # and for 2D the deleted bases come out of the exon (reference) budget: M(first-half-2)
    m_b = np.where(ins, first - half - 1, np.where(dele, first - half - 2, 0))
    ops[:, c] = (m_b << 4) | M
    use[:, c] = (ins | dele) & (m_b > 0)
    c += 1
    for k in range(2):
        ops[:, c] = (gap_len[:, k] << 4) | N
        use[:, c] = gap_len[:, k] > 0
        c += 1
        ops[:, c] = (seg_len[:, k + 1] << 4) | M
        use[:, c] = seg_len[:, k + 1] > 0
        c += 1
    ops[:, c] = (clipR << 4) | S
    use[:, c] = clipR > 0
    flag = np.where(rng.random(n_reads) < 0.5, 0, 16).astype(np.uint16)
    mapq = np.full(n_reads, 60, dtype=np.uint8)
    nh = np.ones(n_reads, dtype=np.int32)
    if profile == "c5":
        nhv = np.array([1, 2, 5, 20])
        nh = nhv[np.searchsorted(np.cumsum([0.7, 0.18, 0.08, 0.04]), rng.random(n_reads)).clip(0, 3)].astype(np.int32)
        mq = np.array([0, 1, 60])
        mapq = mq[np.searchsorted(np.cumsum([0.05, 0.05, 0.9]), rng.random(n_reads)).clip(0, 2)].astype(np.uint8)
        r = rng.random(n_reads)
        flag = flag | np.where(r < 0.01, 0x100, 0).astype(np.uint16) | np.where((r >= 0.01) & (r < 0.015), 0x800, 0).astype(np.uint16)
    strand = np.where(spliced, tx["strand"][loc], ord(".")).astype(np.uint8)
    tid = tx["tid"][loc]
    order = np.lexsort((pos, tid))               # stable: ties keep generation order (ends unsorted within a start)
    ncig = use.sum(1)
    ops_s, use_s = ops[order], use[order]
    cig = ops_s[use_s]
    cig_off = np.zeros(n_reads + 1, dtype=np.uint32)
    cig_off[1:] = np.cumsum(ncig[order])
    return dict(tid=tid[order].astype(np.int32), pos=pos[order].astype(np.int32), flag=flag[order], mapq=mapq[order],
                strand=strand[order], nh=nh[order], cig=cig.astype(np.uint32), cig_off=cig_off)


def make_tile(n_files=2, reads_per_file=1_000_000, profile="c2", seed_base=0x71EB0000, n_loci=20000, first_file=0,
              tx=None) -> SoATile:
    tx = tx if tx is not None else make_transcriptome(n_loci)
    parts = [_reads_for_file(tx, reads_per_file, first_file + f, profile, seed_base) for f in range(n_files)]
    file_off = np.zeros(n_files + 1, dtype=np.uint32)
    file_off[1:] = np.cumsum([len(p["tid"]) for p in parts])
    cig_off = np.zeros(int(file_off[-1]) + 1, dtype=np.uint32)
    base = 0
    q = 1
    for p in parts:
        k = len(p["tid"])
        cig_off[q:q + k] = p["cig_off"][1:].astype(np.uint64) + base
        base += int(p["cig_off"][-1])
        q += k

    def cat(name, dt):
        return np.concatenate([p[name] for p in parts]).astype(dt)

    return SoATile(n_files=n_files, file_off=file_off, tbmerged=np.zeros(n_files, dtype=np.uint8),
                   tid=cat("tid", np.int32), pos=cat("pos", np.int32), flag=cat("flag", np.uint16),
                   mapq=cat("mapq", np.uint8), strand=cat("strand", np.uint8), nh=cat("nh", np.int32), cig_off=cig_off,
                   cig=cat("cig", np.uint32))


def collapsed_to_cov_input(tile: SoATile, groups: dict) -> CovInput:
    """What tiecov would read back from the BAM tiebrush wrote for `groups`: the representatives in
    output order with YC = (double)(float)accYC (YC:f tag, tiebrush.cpp:509)."""
    rep = np.asarray(groups["rep"]).astype(np.int64)
    n = len(rep)
    co = tile.cig_off.astype(np.int64)
    ncig = (co[rep + 1] - co[rep])
    cig_off = np.zeros(n + 1, dtype=np.uint32)
    cig_off[1:] = np.cumsum(ncig)
    tot = int(cig_off[-1])
    rec_of = np.repeat(np.arange(n), ncig)
    within = np.arange(tot) - cig_off[:-1].astype(np.int64)[rec_of]
    cig = tile.cig[co[rep][rec_of] + within]
    yc = np.asarray(groups["yc"]).astype(np.float32).astype(np.float64)
    return CovInput(tid=tile.tid[rep], pos=tile.pos[rep], flag=tile.flag[rep], cig_off=cig_off, cig=cig.astype(np.uint32),
                    yc=yc, strand=tile.strand[rep], yx=np.asarray(groups["yx"]).astype(np.int64))


def header_text(n_files=0):
    lines = ["@HD\tVN:1.6\tSO:coordinate"]
    for nm, ln in zip(REF_NAMES, REF_LENS):
        lines.append("@SQ\tSN:%s\tLN:%d" % (nm, ln))
    return "\n".join(lines) + "\n"


def write_bams(tile: SoATile, prefix: str, level: int = 1):
    """Write the tile as real BAM files `<prefix><f>.bam` (SEQ/QUAL '*', QNAME r<file>_<i>) for end-to-end
    runs of the CLI tools.  Pure Python: meant for small tiles."""
    import struct

    from . import bamio
    paths = []
    for f in range(tile.n_files):
        lo, hi = int(tile.file_off[f]), int(tile.file_off[f + 1])
        recs = []
        for i in range(lo, hi):
            cg = tile.cig[int(tile.cig_off[i]):int(tile.cig_off[i + 1])].tolist()
            aux = b""
            if int(tile.nh[i]) != -(2**31):
                aux += b"NHC" + struct.pack("<B", int(tile.nh[i]))
            if tile.strand[i] in (43, 45):
                aux += b"XSA" + bytes([int(tile.strand[i])])
            recs.append(bamio.encode_record(int(tile.tid[i]), int(tile.pos[i]), int(tile.flag[i]), int(tile.mapq[i]), cg,
                                            b"r%d_%d" % (f, i - lo), aux))
        p = "%s%d.bam" % (prefix, f)
        bamio.write_bam(p, header_text(), REF_NAMES, REF_LENS, b"".join(recs), level)
        paths.append(p)
    return paths


def write_bams_fast(tile: SoATile, prefix: str, level: int = 1, threads: int = 0, seq: bool = False):
    """The same files as write_bams, encoded by the host tool (`tbh_tool mkbam`, C++, one file per worker thread): what the
    end-to-end leg of bench.py uses to lay down 32 x 1M-read inputs in seconds."""
    import os
    import shutil
    import subprocess
    import tempfile
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "tbh_tool")
    d = tempfile.mkdtemp(prefix="tbk_soa_", dir=os.path.dirname(os.path.abspath(prefix)) or None)
    try:
        for name, dt in (("file_off", np.uint32), ("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8),
                         ("strand", np.uint8), ("nh", np.int32), ("cig_off", np.uint32), ("cig", np.uint32)):
            np.ascontiguousarray(getattr(tile, name), dtype=dt).tofile(os.path.join(d, name))
        with open(os.path.join(d, "header.txt"), "w") as fh:
            fh.write(header_text())
        # seq: SEQ / QUAL of the query length and an aligner's tag set on every record (about 250 bytes per 100-bp read, like the
        # reference's fixtures) instead of the bare records ('*' SEQ) that keep the files small
        subprocess.run([tool, "mkbam", d, prefix, str(level), str(threads or 0)] + (["seq"] if seq else []), check=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return ["%s%d.bam" % (prefix, f) for f in range(tile.n_files)]
