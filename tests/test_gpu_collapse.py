"""GPU parity: tbk_collapse_tile (HIP) vs the CPU oracle, bit-exact on every output array, plus the
reference's golden BAMs through the SURVEY.md §4.4 normaliser."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, sample_paths, compare_groups_to_golden_bam, tbk_debug

pytestmark = pytest.mark.gpu
KEYS = ("rep", "yc", "yx", "yd", "g_start", "g_end", "rec_group")


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _check(ctx, tile, device=True, **kw):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api
    okw = dict(kw)
    if "strategy" in okw and isinstance(okw["strategy"], str):
        okw["strategy"] = {"cigar": 0, "full": 1, "clip": 2, "exon": 3}[okw["strategy"]]
    want = orc.collapse(tile, want_rec_group=True, **okw)
    got = api.to_numpy(ctx.collapse(api.to_device(tile, "cuda:0") if device else tile, want_rec_group=True, **kw))
    assert got["n_passed"] == want["n_passed"]
    assert got["n_groups"] == want["n_groups"]
    for k in KEYS:
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k
    return got, want


@pytest.mark.parametrize("name", ["t1", "t2"])
def test_golden_samples(ctx, name, bam_loader):
    from tiebrush_amd import soa
    bams = [bam_loader(p) for p in sample_paths(name)]
    tile = soa.tile_from_bams(bams, with_names=True)
    gold = bam_loader(os.path.join(GOLDEN, name, name + ".bam"))
    got, _ = _check(ctx, tile, collapse_same=True)          # 0.0.6 semantics of the goldens == HEAD with -A
    assert compare_groups_to_golden_bam(got, tile, bams, gold) == []
    _check(ctx, tile)                                       # HEAD default
    _check(ctx, tile, device=False)                         # host-pointer mode


@pytest.mark.parametrize("name", ["t1", "t2", "t12"])
@pytest.mark.parametrize("strategy", ["clip", "exon"])
def test_golden_clip_exon(ctx, name, strategy, bam_loader):
    """fixture CIGARs are M/N only, so -P and -E must give the golden BAMs too (the reference-held pin of the
    strategy code of configs 3 and 5: cmpCigarClip tiebrush.cpp:312-332, cmpExons :334-345)"""
    from tiebrush_amd import soa
    if name == "t12":
        paths = [os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2.bam")]
        gold = bam_loader(os.path.join(GOLDEN, "t12.bam"))
    else:
        paths = sample_paths(name)
        gold = bam_loader(os.path.join(GOLDEN, name, name + ".bam"))
    bams = [bam_loader(p) for p in paths]
    tile = soa.tile_from_bams(bams, with_names=True)
    got, _ = _check(ctx, tile, collapse_same=True, strategy=strategy)
    assert compare_groups_to_golden_bam(got, tile, bams, gold) == []
    _check(ctx, tile, strategy=strategy)                    # HEAD default (-A off), window / sort path as the tile size picks


@pytest.mark.parametrize("machine", ["wave", "lane", "literal"])
def test_golden_t2_yd_machines(ctx, machine, bam_loader, monkeypatch):
    """golden t2 (its 64 tail-drop YDs among them) with every chain forced through yd_wave_k / through yd_lane_k / through yd_run_k
    (the literal list machine the other two hand their overflowing lists to: it reads single exons and the exact two-exon shape from
    the item word, everything else from the groups' exon arrays)"""
    from tiebrush_amd import soa
    if machine == "literal":
        tbk_debug(monkeypatch, yd_literal="1")
    else:
        tbk_debug(monkeypatch, yd_wave_min="1" if machine == "wave" else str(1 << 30))
    bams = [bam_loader(p) for p in sample_paths("t2")]
    tile = soa.tile_from_bams(bams, with_names=True)
    gold = bam_loader(os.path.join(GOLDEN, "t2", "t2.bam"))
    got, _ = _check(ctx, tile, collapse_same=True)
    assert compare_groups_to_golden_bam(got, tile, bams, gold) == []
    tbk_debug(monkeypatch, path="window")
    got, _ = _check(ctx, tile)
    assert int(np.asarray(got["yd"]).max()) > 0


@pytest.mark.parametrize("bgrid", ["1", "3"])
def test_yd_chain_buckets_with_several_chains_per_thread(ctx, bgrid, monkeypatch):
    """the chain buckets are filled by a fixed grid that reserves once per block and bucket: with one and three blocks for every
    chain of the tile each thread takes many chains (at config 3's full size: 3.3 M chains for 1024 blocks); a deep synthetic
    tile against the oracle, default path and forced window path"""
    from tiebrush_amd import synth
    tbk_debug(monkeypatch, yd_bgrid=bgrid)
    tile = synth.make_tile(12, 20000, "c3", n_loci=60)
    got, _ = _check(ctx, tile, strategy="clip")
    assert int(np.asarray(got["yd"]).max()) > 0
    tbk_debug(monkeypatch, path="window")
    _check(ctx, tile, strategy="clip")


def test_golden_t12_tbmerged(ctx, bam_loader):
    from tiebrush_amd import soa
    bams = [bam_loader(os.path.join(GOLDEN, "t1", "t1.bam")), bam_loader(os.path.join(GOLDEN, "t2", "t2.bam"))]
    tile = soa.tile_from_bams(bams, with_names=True)
    gold = bam_loader(os.path.join(GOLDEN, "t12.bam"))
    got, _ = _check(ctx, tile, collapse_same=True)
    assert compare_groups_to_golden_bam(got, tile, bams, gold) == []


def test_golden_full_strategy_with_md(ctx, bam_loader):
    from tiebrush_amd import soa
    bams = [bam_loader(p, keep_md=True) for p in sample_paths("t2")[:4]]
    tile = soa.tile_from_bams(bams, with_md=True)
    _check(ctx, tile, strategy="full")


@pytest.mark.parametrize("profile,kw", [
    ("c2", {}),
    ("c3", dict(strategy="clip")),
    ("c5", dict(strategy="exon", max_nh=5, min_qual=1)),
    ("c5", dict(strategy="cigar", keep_secondary=True, keep_supplementary=True)),
    ("c3", dict(strategy="exon")),
])
def test_synthetic(ctx, profile, kw):
    from tiebrush_amd import synth
    tile = synth.make_tile(4, 60000, profile, n_loci=2000)
    _check(ctx, tile, **kw)


def test_many_files(ctx):
    from tiebrush_amd import synth
    tile = synth.make_tile(64, 4000, "c2", n_loci=300)
    _check(ctx, tile)


def _mk(files, tb=None):
    """files: list of lists of (tid,pos,flag,mapq,strand,nh,cigar[(len,op)..][,yc,yx,yd])"""
    from tiebrush_amd import soa
    recs = [r for f in files for r in f]
    n = len(recs)
    fo = np.zeros(len(files) + 1, np.uint32)
    fo[1:] = np.cumsum([len(f) for f in files])
    cigs = [[(l << 4) | o for l, o in r[6]] for r in recs]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum([len(c) for c in cigs])
    t = soa.SoATile(
        n_files=len(files), file_off=fo, tbmerged=np.array(tb if tb else [0] * len(files), np.uint8),
        tid=np.array([r[0] for r in recs], np.int32), pos=np.array([r[1] for r in recs], np.int32),
        flag=np.array([r[2] for r in recs], np.uint16), mapq=np.array([r[3] for r in recs], np.uint8),
        strand=np.array([ord(r[4]) for r in recs], np.uint8), nh=np.array([r[5] for r in recs], np.int32),
        cig_off=off, cig=np.array([x for c in cigs for x in c], np.uint32))
    if tb and any(tb):
        t.yc_in = np.array([r[7] if len(r) > 7 else 0.0 for r in recs], np.float64)
        t.yx_in = np.array([r[8] if len(r) > 8 else 1 for r in recs], np.int64)
        t.yd_in = np.array([r[9] if len(r) > 9 else 0 for r in recs], np.int64)
    return t


M, I, D, N, S, H = 0, 1, 2, 3, 4, 5


def test_tie_break_and_order(ctx):
    """same (start,end,strand) with different CIGARs: n_cigar first, then memcmp of the little-endian words;
    strand order '+' < '-' < '.'; representative decided by the merge order (prefix-max of end per file)."""
    f0 = [
        (0, 100, 0, 60, "+", 1, [(50, M), (100, N), (50, M)]),     # end 300
        (0, 100, 0, 60, "+", 1, [(100, M)]),                       # end 200: its effective key is (100,300) in file 0
        (0, 100, 0, 60, ".", 1, [(100, M)]),
        (0, 100, 0, 60, "-", 1, [(100, M)]),
        (0, 100, 0, 60, "+", 1, [(40, M), (120, N), (40, M)]),     # end 300, 3 ops
        (0, 100, 0, 60, "+", 1, [(256, M)]),                       # memcmp is byte-wise on little-endian words
        (0, 100, 0, 60, "+", 1, [(17, M), (222, D), (17, M)]),
        (0, 100, 0, 60, "+", 1, [(3, S), (100, M)]),               # soft clip: differs under cigar, equal under clip
        (0, 300, 0, 60, "+", 1, [(16, M), (1, I), (16, M)]),
        (0, 300, 0, 60, "+", 1, [(32, M)]),
        (0, 300, 0, 60, "+", 1, [(16, M), (2, D), (14, M)]),       # same exon as 32M
    ]
    f1 = [
        (0, 100, 0, 60, "+", 1, [(100, M)]),                       # pops before file 0's 100M (end 200 < 300)
        (0, 100, 16, 60, "+", 1, [(50, M), (100, N), (50, M)]),
        (0, 100, 0, 60, "+", 1, [(200, M), (1, I)]),               # end 300
        (0, 300, 0, 60, "+", 1, [(32, M)]),
        (1, 5, 0, 60, ".", 1, [(10, M)]),
        (2, 5, 0, 60, ".", 1, [(10, M)]),
    ]
    tile = _mk([f0, f1])
    for strat in ("cigar", "clip", "exon"):
        got, want = _check(ctx, tile, strategy=strat)
    got, want = _check(ctx, tile)
    # the 100M '+' group: file 1's record pops first (its key end is 200, file 0's is held back by the 300)
    k = [i for i in range(got["n_groups"]) if got["g_end"][i] == 200 and got["yc"][i] == 2.0]
    assert len(k) == 1 and got["rep"][k[0]] == len(f0)


def test_filters_and_unmapped(ctx):
    f0 = [
        (0, 10, 4, 0, ".", -(2**31), []),                # unmapped, placed first
        (0, 10, 0, 60, ".", 1, [(50, M)]),
        (0, 10, 0x100, 60, ".", 1, [(50, M)]),           # secondary
        (0, 10, 0x800, 60, ".", 1, [(50, M)]),           # supplementary
        (0, 10, 0, 3, ".", 1, [(50, M)]),                # low mapq
        (0, 10, 0, 60, ".", 7, [(50, M)]),               # NH 7
        (0, 10, 0, 60, ".", -(2**31), [(50, M)]),        # NH absent -> 0 for the filter
        (0, 20, 4, 0, ".", 1, [(50, M)]),                # unmapped in the middle (start=end=0)
        (0, 30, 0, 60, ".", 1, [(50, M)]),
        (-1, -1, 4, 0, ".", 1, []),                      # unmapped tail
    ]
    f1 = [(0, 10, 0, 60, ".", 1, [(50, M)]), (0, 30, 0, 60, ".", 2, [(50, M)])]
    tile = _mk([f0, f1, []])
    _check(ctx, tile)
    _check(ctx, tile, max_nh=5, min_qual=10)
    _check(ctx, tile, keep_secondary=True, keep_supplementary=True)
    _check(ctx, tile, max_nh=0)
    got, _ = _check(ctx, tile, min_qual=61)              # everything filtered
    assert got["n_groups"] == 0


def test_tbmerged_mixed_with_plain(ctx):
    f0 = [(0, 10, 0, 60, "+", 1, [(50, M)], 3.0, 2, 7), (0, 10, 0, 60, "+", 1, [(50, M)], 0.0, 1, 0),
          (0, 40, 0, 60, "-", 1, [(20, M), (100, N), (30, M)], 300.0, 254, 255)]
    f1 = [(0, 10, 0, 60, "+", 1, [(50, M)]), (0, 10, 0, 60, "+", 1, [(50, M)]),
          (0, 40, 0, 60, "-", 1, [(20, M), (100, N), (30, M)])]
    f2 = [(0, 5, 0, 60, "+", 1, [(50, M)]), (0, 10, 0, 60, "+", 1, [(50, M)])]
    _check(ctx, _mk([f0, f1, f2], tb=[1, 0, 0]))
    _check(ctx, _mk([f1, f0, f2], tb=[0, 1, 0]))


def test_empty_and_tiny(ctx):
    tile = _mk([[], []])
    got = ctx.collapse(tile)
    assert got["n_groups"] == 0 and got["n_passed"] == 0
    _check(ctx, _mk([[(0, 0, 0, 60, ".", 1, [(1, M)])]]))


def test_unsorted_input_is_rejected(ctx):
    from tiebrush_amd import api
    tile = _mk([[(0, 100, 0, 60, ".", 1, [(10, M)]), (0, 50, 0, 60, ".", 1, [(10, M)])]])
    with pytest.raises(api.TbkError) as ei:
        ctx.collapse(tile)
    assert ei.value.status == -6


def test_unsupported_options_fail_loudly(ctx):
    from tiebrush_amd import api
    tile = _mk([[(0, 100, 0, 60, ".", 1, [(10, M)])]])
    for kw in (dict(flags_mask=4), dict(keep_unmapped=True)):
        with pytest.raises(api.TbkError) as ei:
            ctx.collapse(tile, **kw)
        assert ei.value.status == -5


def test_yd_stress_spliced(ctx):
    """few loci, many samples: long YD chains with splice variants exercise the list machine incl. the
    mergeRead tail drop"""
    from tiebrush_amd import synth
    tile = synth.make_tile(12, 20000, "c2", n_loci=40)
    got, want = _check(ctx, tile)
    assert want["yd"].max() > 100


def test_store_frac_ordered_accumulation(ctx):
    """--store-frac: YC is a double accumulated in merge order; sums of 1/NH are order-sensitive in the last ulp"""
    from tiebrush_amd import synth
    tile = synth.make_tile(6, 40000, "c5", n_loci=150)
    got, want = _check(ctx, tile, keep_secondary=True, store_frac=True)
    assert np.any(want["yc"] != np.floor(want["yc"]))
    _check(ctx, tile, keep_secondary=True, store_frac=True, strategy="exon", max_nh=20)


def test_fractional_tbmerged_and_collapse_same(ctx, bam_loader):
    """re-collapse of a --store-frac output (fractional carried YC) mixed with plain files, with and without -A"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import soa, synth
    rng = np.random.default_rng(7)
    base = synth.make_tile(3, 20000, "c5", n_loci=100)
    # file 0 plays a TieBrush output: unique keys per position are not required for the arithmetic
    base.tbmerged = np.array([1, 0, 0], np.uint8)
    n = base.n_records
    base.yc_in = np.where(rng.random(n) < 0.5, rng.integers(1, 9, n) / 5.0, rng.integers(1, 300, n).astype(np.float64))
    base.yx_in = rng.integers(1, 12, n).astype(np.int64)
    base.yd_in = rng.integers(0, 400, n).astype(np.int64)
    _check(ctx, base, keep_secondary=True)
    # -A needs names: synthetic names with deliberate repeats inside a file
    names = [b"r%d" % (i // 2) for i in range(n)]
    lens = np.array([len(x) for x in names])
    base.qn_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    base.qn = np.frombuffer(b"".join(names), np.uint8).copy()
    base.qname_hash = np.array([soa.qname_hash64(x, soa.pair_order(int(f))) for x, f in zip(names, base.flag)], np.uint64)
    _check(ctx, base, keep_secondary=True, collapse_same=True)
    _check(ctx, base, keep_secondary=True, collapse_same=True, store_frac=True)


def test_collapse_same_is_decided_on_the_name_bytes(ctx, monkeypatch):
    """-A: "same read" = same QNAME and pairOrder (tiebrush.cpp:422-424).  The 64-bit name hash is only a filter: with the
    hash narrowed to 2 bits (TBK_DEBUG_QHASH_MASK) nearly every pair of reads collides, and the result is still the
    oracle's (which compares the names), with and without paired flags and --store-frac."""
    from tiebrush_amd import soa, synth
    rng = np.random.default_rng(77)
    base = synth.make_tile(3, 4000, "c2", n_loci=12)
    n = base.n_records
    base.flag = (base.flag | rng.choice([0, 0x40, 0x80], n).astype(np.uint16)).astype(np.uint16)
    names = [b"q%d" % int(x) for x in rng.integers(0, 40, n)]                # few names: many true repeats inside a file
    lens = np.array([len(x) for x in names])
    base.qn_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    base.qn = np.frombuffer(b"".join(names), np.uint8).copy()
    base.qname_hash = np.array([soa.qname_hash64(x, soa.pair_order(int(f))) for x, f in zip(names, base.flag)], np.uint64)
    _check(ctx, base, collapse_same=True)
    tbk_debug(monkeypatch, qhash_mask="0x3")
    _check(ctx, base, collapse_same=True)
    _check(ctx, base, collapse_same=True, store_frac=True)
    tbk_debug(monkeypatch, qhash_mask="0x0")                        # every hash equal: the bytes alone decide
    _check(ctx, base, collapse_same=True)
    tbk_debug(monkeypatch, qhash_mask=None)
    # names are required with -A (no silent hash-only mode)
    from tiebrush_amd import api
    nameless = synth.make_tile(2, 500, "c2", n_loci=5)
    nameless.qname_hash = np.zeros(nameless.n_records, np.uint64)
    with pytest.raises(api.TbkError) as e:
        ctx.collapse(nameless, collapse_same=True)
    assert e.value.status == -1


def test_hash_collision_reseed_path(ctx, monkeypatch):
    """the grouping hash is only a sort accelerator: every non-head is verified against the full key, a collision makes
    the host retry with another seed, and four colliding seeds fail loudly — never a wrong group"""
    from tiebrush_amd import api, synth
    # soft-clipped reads give distinct CIGARs with equal (start, end): xS..yS vs yS..xS
    tile = synth.make_tile(3, 60000, "c3", n_loci=20)
    tbk_debug(monkeypatch, hash_mask="0xFFF")      # 12 hash bits: a few colliding pairs, seed dependent
    seen_reseed = False
    for k in range(6):
        t = synth.make_tile(3, 6000, "c3", n_loci=8 + k)
        try:
            _check(ctx, t)
        except api.TbkError as e:
            assert e.status == -8                          # TBK_ECOLLISION after four seeds: loud, not wrong
            continue
        seen_reseed |= "reseeded" in ctx.last_message()
    tbk_debug(monkeypatch, hash_mask="0x3")        # 2 bits: every seed collides
    with pytest.raises(api.TbkError) as ei:
        ctx.collapse(tile)
    assert ei.value.status == -8
    tbk_debug(monkeypatch, hash_mask=None)
    _check(ctx, tile)


def test_1024_files(ctx):
    """config-5 shape: 1024 input files (16-bit file index, 2048 YD lists)"""
    from tiebrush_amd import synth
    tile = synth.make_tile(1024, 600, "c5", n_loci=150)
    _check(ctx, tile, strategy="exon", max_nh=5, min_qual=1)


def test_yd_list_longer_than_a_wave(ctx):
    """reads with 70+ tiny exons: the per-lane node list of yd_wave_k overflows (64 lanes) and the chain is handed to
    the thread-per-chain kernel; both must reproduce the reference list machine"""
    rng = np.random.default_rng(3)
    files = []
    for f in range(2):
        recs = []
        for i in range(40):
            pos = 100 + 3 * i + int(rng.integers(0, 3))
            cig = []
            for e in range(int(rng.integers(66, 90))):
                cig += [(int(rng.integers(2, 5)), M), (int(rng.integers(3, 9)), N)]
            cig.append((4, M))
            recs.append((0, pos, 0, 60, str(rng.choice(["+", "-", "."])), 1, cig))
            if i % 3 == 0:
                recs.append((0, pos, 0, 60, ".", 1, [(50, M)]))
        recs.sort(key=lambda r: (r[0], r[1]))
        files.append(recs)
    got, want = _check(ctx, _mk(files))
    assert want["yd"].max() > 0


def test_deferred_yd_overlaps_tiecov_chain(ctx):
    """defer_yd: rep/yc/yx are final at return, the YD column is completed by finish_yd() while the caller already runs
    the tiecov chain; results identical to the inline path and to the oracle"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth
    tile = synth.make_tile(4, 50000, "c2", n_loci=400)
    want = orc.collapse(tile)
    cw = orc.coverage(synth.collapsed_to_cov_input(tile, want))
    dt = api.to_device(tile, "cuda:0")
    for rep in range(3):                                   # first call learns the arena size (inline), later ones defer
        res = ctx.collapse(dt, defer_yd=True)
        cov = api.to_numpy(ctx.coverage(ctx.groups_to_cov_in(res)))
        ctx.finish_yd()
        got = api.to_numpy(res)
        for k in ("rep", "yc", "yx", "yd", "g_start", "g_end"):
            assert np.array_equal(got[k], want[k]), (k, rep)
        for k in ("iv_tid", "iv_start", "iv_end", "iv_val", "j_start", "j_end", "j_val"):
            assert np.array_equal(cov[k], cw[k]), (k, rep)
    # a new collapse implies finish_yd of the previous one
    r1 = ctx.collapse(dt, defer_yd=True)
    r2 = api.to_numpy(ctx.collapse(dt))
    assert np.array_equal(r2["yd"], want["yd"])
    with pytest.raises(api.TbkError):
        ctx.collapse(tile, defer_yd=True)                  # host-pointer mode cannot defer


@pytest.mark.parametrize("profile,kw,okw", [("c2", dict(), dict()), ("c3", dict(strategy="clip"), dict(strategy=2)),
                                            ("c5", dict(strategy="exon", max_nh=5, min_qual=1), dict(strategy=3, max_nh=5, min_qual=1))])
def test_device_chain_view_from_keys(ctx, profile, kw, okw, monkeypatch):
    """tbk_groups_out.g_key (ABI 4): the tiecov input of the representatives built from the group keys — the CIGAR itself for a single
    M or M N M under the CIGAR / clip strategies, fetched otherwise (indels, three exons, -E) — gives the intervals and
    junctions of the view that fetches every representative, and of the oracle; the key words say what tbk.h says they say"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth
    tile = synth.make_tile(5, 30000, profile, n_loci=500)
    tile.strand = tile.strand.copy()
    tile.strand[::5] = ord(".")
    want = orc.collapse(tile, **okw)
    cw = orc.coverage(synth.collapsed_to_cov_input(tile, want))
    dt = api.to_device(tile, "cuda:0")
    covs = {}
    for want_key in (True, False):
        res = ctx.collapse(dt, want_key=want_key, **kw)
        covs[want_key] = api.to_numpy(ctx.coverage(ctx.groups_to_cov_in(res)))
        if want_key:
            got = api.to_numpy(res)
            key = np.asarray(got["g_key"]).view(np.uint64).reshape(-1, 2)
            rep = np.asarray(got["rep"]).astype(np.int64)
            assert np.array_equal((key[:, 0] >> np.uint64(33)).astype(np.int64) - 1, tile.tid[rep])
            assert np.array_equal(((key[:, 0] >> np.uint64(2)) & np.uint64(0x7FFFFFFF)).astype(np.int64), np.asarray(got["g_start"]))
            assert np.array_equal((key[:, 1] >> np.uint64(32)).astype(np.int64), np.asarray(got["g_end"]) - np.asarray(got["g_start"]) + 1)
            code = (key[:, 0] & np.uint64(3)).astype(np.int64)
            assert np.array_equal(np.array([ord("+"), ord("-"), ord(".")])[code], tile.strand[rep])
            shape = (key[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
            if kw.get("strategy") == "exon":
                assert not shape.any()                       # exon codes / hashed words say nothing about the CIGAR
            else:
                assert (shape != 0).mean() > 0.5              # most reads are one M or M N M
                ncig = (tile.cig_off[rep + 1] - tile.cig_off[rep]).astype(np.int64)
                one = shape == 0x80000000
                assert (ncig[one] <= 3).all() and (ncig[(shape >> 30) == 3] >= 3).all()
    # the view built from keys carries the results of tiecov's first pass (TBK_COV_PREP: run that pass anyway)
    tbk_debug(monkeypatch, cov_prep="1")
    res = ctx.collapse(dt, want_key=True, **kw)
    covs["pass"] = api.to_numpy(ctx.coverage(ctx.groups_to_cov_in(res)))
    for k in ("iv_tid", "iv_start", "iv_end", "iv_val", "j_tid", "j_start", "j_end", "j_strand", "j_val"):
        assert np.array_equal(covs[True][k], covs[False][k]), k
        assert np.array_equal(covs[True][k], covs["pass"][k]), k
        assert np.array_equal(covs[True][k], cw[k]), k
    for k in ("n_bases", "n_intervals", "n_junctions", "span_bases"):
        assert covs[True][k] == covs[False][k] == covs["pass"][k] == cw[k], k


def _degenerate_exon_files():
    """A CIGAR that ends in an intron (… 7N 2S) leaves a last exon (end + 1, end) in the sample's segment list
    (GSam.cpp:351-417).  The next read of that sample starts exactly at end + 1: processRead does not clear that node
    (its start is not < the read's start), mergeRead then finds no overlap and DROPS the read's exons (tiebrush.cpp:214-216),
    and the read after it measures its distance against what is left.  The list is only renewed by a start beyond end + 1."""
    M, N, S = 0, 3, 4
    f0 = [(2, 9, 0, 60, ".", 1, [(24, M), (7, N), (2, S)]),          # start 10, end 40, exons (10,33) and (41,40)
          (2, 40, 0, 60, ".", 1, [(15, M), (40, N), (10, M)]),       # start 41 = end + 1: its exons are dropped
          (2, 41, 0, 60, "+", 1, [(10, M), (20, N), (30, M)]),       # start 42: d = 0 (a chain cut at 41 would give 1)
          (2, 43, 0, 60, "+", 1, [(30, M)])]
    f1 = [(2, 41, 16, 60, "+", 1, [(10, M), (20, N), (30, M)]), (2, 200, 0, 60, "-", 1, [(30, M)])]
    return [f0, f1]


def test_yd_list_survives_a_start_at_end_plus_one(ctx):
    from test_gpu_window import _tile
    _check(ctx, _tile(_degenerate_exon_files()))
    _check(ctx, _tile(list(reversed(_degenerate_exon_files()))))


@pytest.mark.parametrize("profile,strategy,kw", [("c3", "clip", {}), ("c5", "exon", dict(max_nh=5, min_qual=1)), ("c2", "cigar", {})])
def test_literal_yd_machine_on_synthetic_tiles(ctx, profile, strategy, kw, monkeypatch):
    """every chain through yd_run_k on spliced synthetic tiles (window path forced: items placed by list, their exon words; and the
    default path), against the oracle"""
    from tiebrush_amd import synth
    tbk_debug(monkeypatch, yd_literal="1")
    tile = synth.make_tile(10, 8000, profile, n_loci=60)
    got, _ = _check(ctx, tile, strategy=strategy, **kw)
    assert int(np.asarray(got["yd"]).max()) > 0
    tbk_debug(monkeypatch, yd_literal="1", path="window")
    _check(ctx, tile, strategy=strategy, **kw)
