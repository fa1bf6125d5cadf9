"""Structure-of-arrays tiles that cross the C-ABI (`include/tbk.h`).

`SoATile` mirrors `tbk_soa_in` (file-major concatenation of the decoded records of
every input file); `CovInput` mirrors `tbk_cov_in`.  Arrays are numpy on the host;
`tiebrush_amd.api` moves them to HBM.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

NH_ABSENT = -(2**31)

_FNV_OFF = np.uint64(0xCBF29CE484222325)
_FNV_PRIME = np.uint64(0x100000001B3)


def qname_hash64(name: bytes, pair_order: int) -> int:
    """FNV-1a over the name bytes and GSamRecord::pairOrder (GSam.h:314-320)."""
    h = 0xCBF29CE484222325
    for b in name:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    h = ((h ^ (pair_order + 1)) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def pair_order(flag: int) -> int:
    return 1 if flag & 0x40 else (2 if flag & 0x80 else 0)


@dataclass
class SoATile:
    n_files: int
    file_off: np.ndarray          # uint32 [n_files+1]
    tbmerged: np.ndarray          # uint8  [n_files]
    tid: np.ndarray               # int32
    pos: np.ndarray               # int32
    flag: np.ndarray              # uint16
    mapq: np.ndarray              # uint8
    strand: np.ndarray            # uint8 ASCII
    nh: np.ndarray                # int32, NH_ABSENT when missing
    cig_off: np.ndarray           # uint32 [n+1]
    cig: np.ndarray               # uint32
    yc_in: Optional[np.ndarray] = None   # float64
    yx_in: Optional[np.ndarray] = None   # int64
    yd_in: Optional[np.ndarray] = None   # int64
    md_off: Optional[np.ndarray] = None  # uint32 [n+1]
    md: Optional[np.ndarray] = None      # uint8
    md_has: Optional[np.ndarray] = None  # uint8
    qname_hash: Optional[np.ndarray] = None  # uint64
    qn_off: Optional[np.ndarray] = None  # uint32 [n+1]   (oracle only: -A compares real names)
    qn: Optional[np.ndarray] = None      # uint8
    prio_hi: Optional[np.ndarray] = None  # uint64: explicit merge-order priority (cross-rank stitch only)
    prio_lo: Optional[np.ndarray] = None  # uint64

    @property
    def n_records(self) -> int:
        return int(self.tid.shape[0])

    def file_of(self) -> np.ndarray:
        return (np.searchsorted(self.file_off, np.arange(self.n_records), side="right") - 1).astype(np.int32)

    def validate(self):
        n = self.n_records
        assert self.file_off.dtype == np.uint32 and self.file_off.shape == (self.n_files + 1,)
        assert int(self.file_off[0]) == 0 and int(self.file_off[-1]) == n
        assert self.cig_off.shape == (n + 1,) and int(self.cig_off[-1]) == self.cig.shape[0]
        for a, dt in ((self.tid, np.int32), (self.pos, np.int32), (self.flag, np.uint16), (self.mapq, np.uint8),
                      (self.strand, np.uint8), (self.nh, np.int32)):
            assert a.dtype == dt and a.shape == (n,), (a.dtype, dt, a.shape)


def tile_from_bams(bams, with_names: bool = False, with_md: bool = False) -> SoATile:
    """Concatenate decoded BAM files (`bamio.BamSoA`) file-major into one tile."""
    k = len(bams)
    file_off = np.zeros(k + 1, dtype=np.uint32)
    file_off[1:] = np.cumsum([b.n for b in bams])
    tb = np.array([1 if b.header.is_tiebrush() else 0 for b in bams], dtype=np.uint8)

    def cat(name, dt):
        return np.concatenate([getattr(b, name) for b in bams]).astype(dt) if k else np.zeros(0, dt)

    cig_off = np.zeros(int(file_off[-1]) + 1, dtype=np.uint32)
    base = 0
    p = 1
    for b in bams:
        cig_off[p:p + b.n] = b.cig_off[1:].astype(np.uint64) + base
        base += int(b.cig_off[-1])
        p += b.n
    nh = cat("nh", np.int64)
    t = SoATile(
        n_files=k, file_off=file_off, tbmerged=tb, tid=cat("tid", np.int32), pos=cat("pos", np.int32),
        flag=cat("flag", np.uint16), mapq=cat("mapq", np.uint8), strand=cat("strand", np.uint8),
        nh=nh.astype(np.int32), cig_off=cig_off, cig=cat("cig", np.uint32))
    if tb.any():
        t.yc_in = cat("yc", np.float64)
        t.yx_in = cat("yx", np.int64)
        t.yd_in = cat("yd", np.int64)
    if with_names:
        names = [q for b in bams for q in b.qname]
        lens = np.fromiter((len(q) for q in names), dtype=np.int64, count=len(names))
        t.qn_off = np.zeros(len(names) + 1, dtype=np.uint32)
        t.qn_off[1:] = np.cumsum(lens)
        t.qn = np.frombuffer(b"".join(names), dtype=np.uint8).copy()
        fl = t.flag.tolist()
        t.qname_hash = np.fromiter((qname_hash64(q, pair_order(f)) for q, f in zip(names, fl)),
                                   dtype=np.uint64, count=len(names))
    if with_md:
        mds = [m for b in bams for m in b.md]
        lens = np.fromiter((0 if m is None else len(m) for m in mds), dtype=np.int64, count=len(mds))
        t.md_off = np.zeros(len(mds) + 1, dtype=np.uint32)
        t.md_off[1:] = np.cumsum(lens)
        t.md = np.frombuffer(b"".join(m for m in mds if m is not None), dtype=np.uint8).copy()
        t.md_has = np.fromiter((0 if m is None else 1 for m in mds), dtype=np.uint8, count=len(mds))
    return t


@dataclass
class CovInput:
    tid: np.ndarray      # int32
    pos: np.ndarray      # int32
    flag: np.ndarray     # uint16
    cig_off: np.ndarray  # uint32 [n+1]
    cig: np.ndarray      # uint32
    yc: np.ndarray       # float64 (1.0 when the tag is absent)
    strand: Optional[np.ndarray] = None  # uint8
    yx: Optional[np.ndarray] = None      # int64

    @property
    def n_records(self) -> int:
        return int(self.tid.shape[0])


def cov_input_from_bam(b) -> CovInput:
    yc = np.where(b.has_yc, b.yc, 1.0).astype(np.float64)   # tiecov.cpp:482-485
    return CovInput(tid=b.tid.astype(np.int32), pos=b.pos.astype(np.int32), flag=b.flag.astype(np.uint16),
                    cig_off=b.cig_off.astype(np.uint32), cig=b.cig.astype(np.uint32), yc=yc,
                    strand=b.strand.astype(np.uint8), yx=b.yx.astype(np.int64))


@dataclass
class PackedTile:
    """mirror of tbk_packed_in (include/tbk.h): the wire form of a plain tile — 9 bytes per record + the CIGAR words"""
    n_files: int
    file_off: np.ndarray       # uint32 [n_files + 1]
    tid_run_end: np.ndarray    # uint32
    tid_run_tid: np.ndarray    # int32
    pos: np.ndarray            # int32
    meta: np.ndarray           # uint32: flag : 12 | strand code : 2 | mapq : 8 | NH code : 10
    ncig: np.ndarray           # uint8
    cig: np.ndarray            # uint32
    nh_esc_idx: np.ndarray
    nh_esc_val: np.ndarray
    ncig_esc_idx: np.ndarray
    ncig_esc_val: np.ndarray

    @property
    def n_records(self) -> int:
        return int(self.pos.shape[0])

    def nbytes(self) -> int:
        return sum(int(a.nbytes) for a in (self.pos, self.meta, self.ncig, self.cig, self.tid_run_end, self.tid_run_tid, self.nh_esc_idx,
                                           self.nh_esc_val, self.ncig_esc_idx, self.ncig_esc_val))


def pack_tile(tile: SoATile) -> PackedTile:
    """host-side packer (numpy restatement of what a decoder emits): refuses what the wire form does not carry"""
    if np.any(tile.tbmerged) or tile.md_off is not None or tile.prio_hi is not None:
        raise ValueError("only plain tiles have a packed form")
    flag = np.asarray(tile.flag).astype(np.uint32)
    if np.any(flag >> 12):
        raise ValueError("flag bits beyond 0xFFF")
    st = np.asarray(tile.strand)
    sc = np.where(st == ord("+"), 0, np.where(st == ord("-"), 1, 2)).astype(np.uint32)
    if np.any((st != ord("+")) & (st != ord("-")) & (st != ord("."))):
        raise ValueError("strand other than + - .")
    nh = np.asarray(tile.nh).astype(np.int64)
    absent = nh == NH_ABSENT
    esc = ~absent & ((nh < 0) | (nh > 1021))
    code = np.where(absent, 1022, np.where(esc, 1023, nh)).astype(np.uint32)
    meta = flag | (sc << 12) | (np.asarray(tile.mapq).astype(np.uint32) << 14) | (code << 22)
    cnt = np.diff(np.asarray(tile.cig_off).astype(np.int64))
    big = cnt >= 255
    tid = np.asarray(tile.tid)
    n = tile.n_records
    # runs of equal tid; a run never spans two files
    brk = np.ones(n, bool)
    if n:
        brk[1:] = tid[1:] != tid[:-1]
        brk[np.asarray(tile.file_off[:-1])[np.asarray(tile.file_off[:-1]) < n]] = True
    starts = np.nonzero(brk)[0]
    run_end = np.concatenate([starts[1:], [n]]).astype(np.uint32) if n else np.array([0], np.uint32)
    run_tid = tid[starts].astype(np.int32) if n else np.array([-1], np.int32)
    return PackedTile(n_files=tile.n_files, file_off=np.asarray(tile.file_off, np.uint32), tid_run_end=run_end, tid_run_tid=run_tid,
                      pos=np.ascontiguousarray(tile.pos, np.int32), meta=meta.astype(np.uint32), ncig=np.where(big, 255, cnt).astype(np.uint8),
                      cig=np.ascontiguousarray(tile.cig, np.uint32), nh_esc_idx=np.nonzero(esc)[0].astype(np.uint32),
                      nh_esc_val=nh[esc].astype(np.int32), ncig_esc_idx=np.nonzero(big)[0].astype(np.uint32), ncig_esc_val=cnt[big].astype(np.uint32))
