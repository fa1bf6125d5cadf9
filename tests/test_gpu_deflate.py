"""Device-side BGZF deflate (tbk_bgzf_deflate, bgzdef.hip) against zlib: every member the kernel writes must be a well-formed BGZF
member (gzip header, BC field, BSIZE), inflate with zlib to exactly its payload, and carry the payload's CRC32 and ISIZE; the run
must read back through python's gzip, through the repo's host codec and through the device inflate.  Sizes are compared with zlib
level 6 (what htslib's writer uses, GSam.h:648-653)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def members(run: bytes):
    """(deflate bytes, crc, isize) of every member of a BGZF run; asserts the framing"""
    out, o = [], 0
    while o < len(run):
        assert run[o:o + 4] == b"\x1f\x8b\x08\x04" and run[o + 10:o + 12] == b"\x06\x00" and run[o + 12:o + 16] == b"BC\x02\x00", o
        bsize = struct.unpack_from("<H", run, o + 16)[0] + 1
        assert bsize <= 65536 and o + bsize <= len(run)
        crc, isize = struct.unpack_from("<II", run, o + bsize - 8)
        out.append((run[o + 18:o + bsize - 8], crc, isize))
        o += bsize
    assert o == len(run)
    return out


def check_run(run: bytes, payload: bytes, cuts=None):
    ms = members(run)
    o = 0
    for k, (z, crc, isize) in enumerate(ms):
        d = zlib.decompressobj(-15)
        got = d.decompress(z) + d.flush()
        assert d.eof and not d.unused_data, k                      # exactly one finished deflate stream, nothing behind it
        assert len(got) == isize and got == payload[o:o + isize], (k, o, isize)
        assert zlib.crc32(got) & 0xFFFFFFFF == crc, k
        assert isize <= 0xff00
        if cuts is not None:
            pass
        o += isize
    assert o == len(payload)
    return ms


def _payloads():
    rng = np.random.default_rng(3)
    text = (b"ACGTTGCA" * 40 + b"read_%d\tchr1\t100M\n") * 3000
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    skew = np.repeat(np.arange(24, dtype=np.uint8), fib)           # byte frequencies 1, 1, 2, 3, 5, ...: code lengths beyond 15 without the limit
    rng.shuffle(skew)
    far = b"".join(blk + rng.integers(0, 256, gap, dtype=np.uint8).tobytes() + blk
                   for blk, gap in ((rng.integers(0, 256, 700, dtype=np.uint8).tobytes(), g) for g in (8100, 9000, 15000, 31000, 32767 - 700, 32768 - 700, 32769 - 700, 40000)))
    return {
        "random": rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),      # incompressible: stored blocks
        "text": text,                                                            # long matches, overlapping copies
        "runs": b"\0" * 70_000 + b"\xff" * 70_000 + bytes(range(256)) * 300,     # distance-1 copies of the maximum length
        "mixed": rng.integers(0, 4, 200_000, dtype=np.uint8).tobytes() + text[:100_000],
        "far": far,                                                              # matches right at deflate's 32 KiB reach, and beyond it
        "skew": skew.tobytes(),                                                  # the length limit of the literal code
        "two_symbols": bytes(rng.integers(0, 2, 70_000, dtype=np.uint8)),
        "tiny": b"x",
        "three": b"abc",
        "exact_member": rng.integers(0, 16, 0xff00, dtype=np.uint8).tobytes(),
        "member_plus_one": rng.integers(0, 16, 0xff00 + 1, dtype=np.uint8).tobytes(),
        "lengths": b"".join(bytes([65 + (k % 26)]) * k + b"#" for k in range(1, 300)),   # runs of every length around 258
    }


@pytest.mark.parametrize("name", sorted(_payloads()))
def test_members_inflate_with_zlib(ctx, name):
    payload = _payloads()[name]
    run = ctx.bgzf_deflate(payload)
    ms = check_run(run, payload)
    assert len(ms) == (len(payload) + 0xff00 - 1) // 0xff00
    assert gzip.decompress(run) == payload                                       # python's multi-member gzip reader
    assert ctx.bgzf_inflate(run) == payload                                      # the device inflate (bamdev.hip)
    zl = sum(len(zlib.compress(payload[o:o + 0xff00], 6)) for o in range(0, len(payload), 0xff00))
    if name == "random":
        assert len(run) <= len(payload) + len(ms) * (18 + 8 + 5)                 # stored, never expanded further
    if name in ("text", "runs", "mixed", "two_symbols", "lengths"):
        assert len(run) <= 1.25 * zl + 512 * len(ms), (name, len(run), zl)     # (runs a kilobyte small are all header)


def test_empty_run(ctx):
    assert ctx.bgzf_deflate(b"") == b""


def test_caller_cuts_including_empty_and_tiny_members(ctx):
    rng = np.random.default_rng(11)
    payload = (b"@SQ\tSN:chr1\tLN:1000\n" * 2000 + rng.integers(0, 4, 90_000, dtype=np.uint8).tobytes())
    cuts = [0, 0, 1, 5, 5, 70, 4000, 4000 + 0xff00, len(payload) - 3, len(payload)]
    cuts = sorted(set(c for c in cuts if c <= len(payload))) + []
    # (pieces above 0xff00 are refused: split the long one)
    full = [0]
    for c in cuts[1:]:
        while c - full[-1] > 0xff00:
            full.append(full[-1] + 0xff00)
        full.append(c)
    full.insert(1, 0)                                       # an empty member: nothing is written for it
    run = ctx.bgzf_deflate(payload, cuts=full)
    ms = check_run(run, payload)
    assert [m[2] for m in ms] == [b - a for a, b in zip(full[:-1], full[1:]) if b > a]
    with pytest.raises(Exception):
        ctx.bgzf_deflate(payload, cuts=[0, len(payload)])  # one piece beyond 0xff00


@pytest.mark.parametrize("name", ["t1/t1s0.bam", "t2/t2.bam", "t12.bam"])
def test_reference_records_deflate_near_level_6(ctx, name):
    """the reference's own records (SEQ, QUAL, an aligner's tags): every member exact, the run within 10 % of zlib level 6"""
    payload = gzip.open(os.path.join(GOLDEN, name)).read()
    run = ctx.bgzf_deflate(payload)
    check_run(run, payload)
    z6 = sum(len(zlib.compress(payload[o:o + 0xff00], 6)) - 6 + 26 for o in range(0, len(payload), 0xff00))   # (zlib wrapper off, BGZF framing on)
    ratio = len(run) / z6
    print("%s: %d -> %d bytes on the device, %d at zlib level 6: %.3f" % (name, len(payload), len(run), z6, ratio))
    assert ratio <= 1.10, ratio


def test_fuzz_small_payloads(ctx):
    rng = np.random.default_rng(17)
    for k in range(60):
        n = int(rng.integers(1, 5000))
        alpha = int(rng.choice([1, 2, 4, 16, 256]))
        p = rng.integers(0, alpha, n, dtype=np.uint8).tobytes()
        if k % 3 == 0:
            p = p[:max(1, n // 7)] * 7                         # repeats at one distance
        run = ctx.bgzf_deflate(p)
        check_run(run, p)


def test_large_run(ctx):
    """a few hundred members in one launch (more members than resident workgroups): the persistent blocks take them all"""
    raw = gzip.open(os.path.join(GOLDEN, "t1/t1s0.bam")).read()
    payload = (raw * 4)[:48_000_000]
    run = ctx.bgzf_deflate(payload)
    check_run(run, payload)
