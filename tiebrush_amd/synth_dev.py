"""Synthetic sorted-BAM workloads generated with torch ops, on whatever device is asked for.

Same model as `synth.py` (shared transcriptome, Zipf(1.1)-weighted locus choice, 100-bp reads, the c2 / c3 / c5
profiles) but every per-read step is a tensor op, so config 3 at its full 64 x 5M reads is built on the GPU in seconds
instead of minutes of numpy on the host.  The random stream is a counter-based SplitMix64 (seed 0x71EB0000 + file index,
one 64-bit draw per (read, purpose)), computed in wrapping int64 arithmetic — the same tile comes out on CPU and GPU.

The arrays of the returned SoATile are torch tensors on `device` (what `api.Context.collapse` takes as
TBK_MEM_DEVICE input); `tile_to_host` brings a tile (or a coordinate window of it) back as numpy for the oracle.
"""
from __future__ import annotations

import numpy as np

from . import synth
from .soa import SoATile

M, I, D, N, S = 0, 1, 2, 3, 4
READ_LEN = synth.READ_LEN
MAX_EXONS = synth.MAX_EXONS


def _i64(x):
    """Python int -> the int64 with the same 64 bits."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


_GAMMA = _i64(0x9E3779B97F4A7C15)
_M1 = _i64(0xBF58476D1CE4E5B9)
_M2 = _i64(0x94D049BB133111EB)


def _mix_py(z):
    """SplitMix64 finaliser on a Python int"""
    z &= (1 << 64) - 1
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
    return z ^ (z >> 31)


def _lsr(x, k):
    """logical shift right of int64 tensors"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _splitmix(torch, ctr, seed, stream):
    """64 random bits per element of the int64 counter tensor (SplitMix64 finaliser of seed + stream + counter)."""
    z = ctr * _GAMMA + _i64(_mix_py(_mix_py(seed) + stream * 0xD1B54A32D192ED03))   # hashed base: files / purposes are unrelated streams
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _u53(torch, r):
    """uniform double in [0,1) from 64 random bits (exact on every device: 53-bit integer * 2^-53)"""
    return _lsr(r, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def _below(torch, r, n):
    """uniform integer in [0, n) for int64 tensor / scalar n <= 2^31: (32 random bits * n) >> 32"""
    return (_lsr(r, 32) * n) >> 32


def _reads_for_file(torch, txd, n_reads, fidx, profile, seed_base, device):
    ctr = torch.arange(n_reads, dtype=torch.int64, device=device)
    seed = seed_base + fidx

    def rnd(stream):
        return _splitmix(torch, ctr, seed, stream)

    loc = torch.searchsorted(txd["cdf"], _u53(torch, rnd(1))).clamp_(0, txd["cdf"].numel() - 1)
    tlen = txd["tlen"][loc]
    clipL = torch.zeros(n_reads, dtype=torch.int64, device=device)
    clipR = torch.zeros(n_reads, dtype=torch.int64, device=device)
    if profile == "c3":
        sc = _u53(torch, rnd(2)) < 0.10
        clipL = torch.where(sc, _below(torch, rnd(3), 9), clipL)
        clipR = torch.where(sc, _below(torch, rnd(4), 9), clipR)
    mlen = READ_LEN - clipL - clipR
    off = _below(torch, rnd(5), tlen - mlen + 1)
    t = txd["t"][loc]                                   # [n, 9]
    e0 = (off[:, None] >= t[:, 1:]).sum(1)
    ex_len = txd["ex_len"][loc]
    g = txd["g"][loc]
    rows = ctr
    pos = g[rows, e0] + (off - t[rows, e0])
    rem = mlen.clone()
    seg_len = torch.zeros((n_reads, 3), dtype=torch.int64, device=device)
    gap_len = torch.zeros((n_reads, 2), dtype=torch.int64, device=device)
    avail = t[rows, e0 + 1] - off
    e = e0.clone()
    for k in range(3):
        take = torch.minimum(rem, avail)
        seg_len[:, k] = take
        rem = rem - take
        more = rem > 0
        if k < 2:
            en = torch.clamp(e + 1, max=MAX_EXONS - 1)
            gap = g[rows, en] - (g[rows, e] + ex_len[rows, e])
            gap_len[:, k] = torch.where(more, gap, torch.zeros_like(gap))
            avail = torch.where(more, ex_len[rows, en], torch.zeros_like(avail))
            e = torch.where(more, en, e)
    clipR = clipR + rem
    spliced = gap_len[:, 0] > 0
    ins = torch.zeros(n_reads, dtype=torch.bool, device=device)
    dele = torch.zeros(n_reads, dtype=torch.bool, device=device)
    if profile == "c5":
        r = _u53(torch, rnd(6))
        ok = seg_len[:, 0] >= 40
        ins = ok & (r < 0.015)
        dele = ok & (r >= 0.015) & (r < 0.03)
    ops = torch.zeros((n_reads, 12), dtype=torch.int64, device=device)
    use = torch.zeros((n_reads, 12), dtype=torch.bool, device=device)
    c = 0
    ops[:, c] = (clipL << 4) | S
    use[:, c] = clipL > 0
    c += 1
    first = seg_len[:, 0]
    half = first // 2
    m_a = torch.where(ins | dele, half, first)
    ops[:, c] = (m_a << 4) | M
    use[:, c] = m_a > 0
    c += 1
    ops[:, c] = torch.where(ins, torch.full_like(first, (1 << 4) | I), torch.full_like(first, (2 << 4) | D))
    use[:, c] = ins | dele
    c += 1
    m_b = torch.where(ins, first - half - 1, torch.where(dele, first - half - 2, torch.zeros_like(first)))
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
    flag = torch.where(_u53(torch, rnd(7)) < 0.5, 0, 16).to(torch.int16)
    mapq = torch.full((n_reads,), 60, dtype=torch.uint8, device=device)
    nh = torch.ones(n_reads, dtype=torch.int32, device=device)
    if profile == "c5":
        nhv = torch.tensor([1, 2, 5, 20], dtype=torch.int32, device=device)
        cd = torch.tensor(np.cumsum([0.7, 0.18, 0.08, 0.04]), dtype=torch.float64, device=device)
        nh = nhv[torch.searchsorted(cd, _u53(torch, rnd(8))).clamp_(0, 3)]
        mq = torch.tensor([0, 1, 60], dtype=torch.uint8, device=device)
        cq = torch.tensor(np.cumsum([0.05, 0.05, 0.9]), dtype=torch.float64, device=device)
        mapq = mq[torch.searchsorted(cq, _u53(torch, rnd(9))).clamp_(0, 2)]
        r = _u53(torch, rnd(10))
        flag = flag | torch.where(r < 0.01, 0x100, 0).to(torch.int16) | torch.where((r >= 0.01) & (r < 0.015), 0x800, 0).to(torch.int16)
    strand = torch.where(spliced, txd["strand"][loc], torch.full((n_reads,), ord("."), dtype=torch.uint8, device=device))
    tid = txd["tid"][loc]
    key = (tid.to(torch.int64) << 32) | pos
    order = torch.sort(key, stable=True)[1]            # ties keep generation order (ends unsorted within a start)
    ncig = use.sum(1)
    ops_s, use_s = ops[order], use[order]
    cig = ops_s[use_s].to(torch.int32)
    return dict(tid=tid[order].to(torch.int32), pos=pos[order].to(torch.int32), flag=flag[order], mapq=mapq[order],
                strand=strand[order], nh=nh[order], cig=cig, ncig=ncig[order])


def make_tile_device(n_files=2, reads_per_file=1_000_000, profile="c2", device="cuda:0", seed_base=0x71EB0000, n_loci=20000,
                     first_file=0, tx=None, chunk=2_500_000) -> SoATile:
    """File-major SoA tile whose per-record arrays are torch tensors on `device`."""
    import torch
    tx = tx if tx is not None else synth.make_transcriptome(n_loci)
    txd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in tx.items()
           if k in ("tid", "ex_len", "g", "t", "strand", "tlen", "cdf")}
    txd["tid"] = txd["tid"].to(torch.int32)
    n = n_files * reads_per_file
    if n >= 2**32:
        raise ValueError("tile too large: %d records" % n)
    out = dict(tid=torch.empty(n, dtype=torch.int32, device=device), pos=torch.empty(n, dtype=torch.int32, device=device),
               flag=torch.empty(n, dtype=torch.int16, device=device), mapq=torch.empty(n, dtype=torch.uint8, device=device),
               strand=torch.empty(n, dtype=torch.uint8, device=device), nh=torch.empty(n, dtype=torch.int32, device=device))
    ncig_all = torch.empty(n, dtype=torch.int32, device=device)
    cigs = []
    for f in range(n_files):
        p = _reads_for_file(torch, txd, reads_per_file, first_file + f, profile, seed_base, device)
        lo = f * reads_per_file
        for k in out:
            out[k][lo:lo + reads_per_file] = p[k]
        ncig_all[lo:lo + reads_per_file] = p["ncig"].to(torch.int32)
        cigs.append(p["cig"])
        del p
    cig = torch.cat(cigs)
    del cigs
    if cig.numel() >= 2**32:
        raise ValueError("tile too large: %d CIGAR ops" % cig.numel())
    cig_off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(ncig_all, 0, out=cig_off[1:])
    cig_off = cig_off.to(torch.int32)                  # uint32 bit pattern (wraps above 2^31, as the C ABI reads it)
    file_off = (np.arange(n_files + 1, dtype=np.uint64) * reads_per_file).astype(np.uint32)
    return SoATile(n_files=n_files, file_off=file_off, tbmerged=np.zeros(n_files, dtype=np.uint8), tid=out["tid"], pos=out["pos"],
                   flag=out["flag"], mapq=out["mapq"], strand=out["strand"], nh=out["nh"], cig_off=cig_off, cig=cig)


def tile_to_host(tile: SoATile, window=None) -> SoATile:
    """numpy copy of a device tile.  `window` = (tid, pos_lo, pos_hi) keeps, for every file, only the records of that
    reference sequence with pos_lo <= pos < pos_hi — a contiguous range of each file, so the sub-tile is what the same
    inputs restricted to that region would be (same depth, same duplication): the bounded CPU-baseline sample."""
    import torch
    n_files = tile.n_files
    fo = np.asarray(tile.file_off, dtype=np.int64)
    if window is None:
        sel = [(int(fo[f]), int(fo[f + 1])) for f in range(n_files)]
    else:
        wt, lo, hi = window
        key = (tile.tid.to(torch.int64) << 32) | tile.pos.to(torch.int64)
        klo, khi = (wt << 32) | lo, (wt << 32) | hi
        sel = []
        for f in range(n_files):
            a, b = int(fo[f]), int(fo[f + 1])
            kk = key[a:b]
            q = torch.searchsorted(kk, torch.tensor([klo, khi], dtype=torch.int64, device=kk.device))
            sel.append((a + int(q[0]), a + int(q[1])))
    co = tile.cig_off.to(torch.int64) & 0xFFFFFFFF

    def cat(t, dt):
        return np.concatenate([t[a:b].cpu().numpy() for a, b in sel]).view(dt) if sel else np.zeros(0, dt)

    new_off = np.zeros(n_files + 1, dtype=np.uint32)
    new_off[1:] = np.cumsum([b - a for a, b in sel])
    ncig = np.concatenate([(co[a + 1:b + 1] - co[a:b]).cpu().numpy() for a, b in sel])
    cig = np.concatenate([tile.cig[int(co[a]):int(co[b])].cpu().numpy() for a, b in sel]).view(np.uint32)
    cig_off = np.zeros(int(new_off[-1]) + 1, dtype=np.uint32)
    cig_off[1:] = np.cumsum(ncig)
    return SoATile(n_files=n_files, file_off=new_off, tbmerged=np.asarray(tile.tbmerged).copy(), tid=cat(tile.tid, np.int32),
                   pos=cat(tile.pos, np.int32), flag=cat(tile.flag, np.uint16), mapq=cat(tile.mapq, np.uint8),
                   strand=cat(tile.strand, np.uint8), nh=cat(tile.nh, np.int32), cig_off=cig_off, cig=cig)
