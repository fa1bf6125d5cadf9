"""GPU parity: tbk_coverage_tile (HIP) vs the CPU oracle, bit-exact (integers / exact doubles)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, bedgraph_lines, junction_lines, read_lines, tbk_debug

pytestmark = pytest.mark.gpu

KEYS = ("iv_tid", "iv_start", "iv_end", "iv_val", "j_tid", "j_start", "j_end", "j_strand", "j_val")


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _check(ctx, cin, device):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api
    want = orc.coverage(cin)
    got = api.to_numpy(ctx.coverage(api.to_device(cin, "cuda:0") if device else cin))
    assert got["n_bases"] == want["n_bases"] and got["span_bases"] == want["span_bases"]
    assert got["n_intervals"] == want["n_intervals"] and got["n_junctions"] == want["n_junctions"]
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
    return got


@pytest.mark.parametrize("name", ["t1", "t2"])
@pytest.mark.parametrize("device", [False, True])
def test_golden_tiecov(ctx, name, device, bam_loader):
    from tiebrush_amd import soa
    b = bam_loader(os.path.join(GOLDEN, name, name + ".bam"))
    got = _check(ctx, soa.cov_input_from_bam(b), device)
    names = b.header.ref_names
    assert bedgraph_lines(got, names) == read_lines(os.path.join(GOLDEN, name, name + ".coverage.bedgraph"))
    assert junction_lines(got, names) == read_lines(os.path.join(GOLDEN, name, name + ".junctions.bed"))


@pytest.mark.parametrize("bundles", ["lean", "legacy", "scan", "refused", "junc_radix", "junc_overflow"])
@pytest.mark.parametrize("profile,n", [("c2", 100000), ("c3", 60000), ("c5", 60000)])
def test_synthetic_collapsed(ctx, profile, n, bundles, monkeypatch):
    """(interval chain: the lean one — one read-back, compacted starts from the head sums —, the general one behind it
    (TBK_COV_LEGACY; with TBK_COV_BUNDLE_SCAN its bundles come from the two-stage look-back scan), and the lean one refusing an
    input for its tile tables (TBK_COV_TILE_CAP) so that the general chain takes over)"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth
    if bundles == "scan":
        tbk_debug(monkeypatch, cov_bundle_scan="1")
    elif bundles == "legacy":
        tbk_debug(monkeypatch, cov_legacy="1")
    elif bundles == "refused":
        tbk_debug(monkeypatch, cov_tile_cap="3")
    elif bundles == "junc_radix":          # junctions: the global radix sort instead of the per-home LDS sorts
        tbk_debug(monkeypatch, junc_radix="1")
    elif bundles == "junc_overflow":       # ... and a home block with more items than its sort takes: the radix path takes over
        tbk_debug(monkeypatch, jh_cap="2")
    tile = synth.make_tile(3, n, profile, n_loci=3000)
    groups = orc.collapse(tile)
    cin = synth.collapsed_to_cov_input(tile, groups)
    _check(ctx, cin, True)


@pytest.mark.parametrize("chain", ["lean", "legacy"])
def test_uncollapsed_deep(ctx, chain, monkeypatch):
    """raw (uncollapsed) reads: deep pile-ups, YC absent -> 1.0"""
    from tiebrush_amd import synth, soa
    if chain == "legacy":
        tbk_debug(monkeypatch, cov_legacy="1")
    tile = synth.make_tile(1, 200000, "c2", n_loci=50)
    cin = soa.CovInput(tid=tile.tid, pos=tile.pos, flag=tile.flag, cig_off=tile.cig_off, cig=tile.cig,
                       yc=np.ones(tile.n_records), strand=tile.strand)
    _check(ctx, cin, True)


@pytest.mark.parametrize("chain", ["lean", "legacy"])
def test_edge_cases(ctx, chain, monkeypatch):
    from tiebrush_amd import soa
    M, I, D, N, S = 0, 1, 2, 3, 4
    if chain == "legacy":
        tbk_debug(monkeypatch, cov_legacy="1")

    def mk(recs):
        tid = np.array([r[0] for r in recs], np.int32)
        pos = np.array([r[1] for r in recs], np.int32)
        flag = np.array([r[2] for r in recs], np.uint16)
        cigs = [[(l << 4) | o for l, o in r[3]] for r in recs]
        off = np.zeros(len(recs) + 1, np.uint32)
        off[1:] = np.cumsum([len(c) for c in cigs])
        cig = np.array([x for c in cigs for x in c], np.uint32)
        yc = np.array([r[4] for r in recs], np.float64)
        st = np.array([ord(r[5]) for r in recs], np.uint8)
        return soa.CovInput(tid, pos, flag, off, cig, yc, st)

    # adjacent bundles (start == b_end + 1) with equal depth must not merge; unmapped skipped;
    # D inside a read leaves a zero gap; tile boundary crossing at 8192; tid change; I/S ignored
    recs = [
        (0, 10, 0, [(50, M)], 3.0, "."),
        (0, 60, 0, [(50, M)], 3.0, "."),          # adjacent: new bundle, same depth
        (0, 60, 4, [(50, M)], 9.0, "."),          # unmapped: skipped
        (0, 200, 0, [(10, M), (5, D), (10, M)], 2.0, "."),
        (0, 8150, 0, [(3, S), (100, M), (2, S)], 1.0, "."),   # crosses the first tile end in cpos? (cpos small) fine
        (0, 9000, 0, [(30, M), (1000, N), (20, M), (2, I), (30, M)], 4.0, "+"),
        (0, 9000, 16, [(30, M), (1000, N), (50, M)], 2.0, "+"),
        (0, 9010, 0, [(20, M), (1000, N), (50, M)], 1.0, "-"),
        (1, 5, 0, [(20000, M)], 1.0, "."),        # long read spanning several tiles
        (1, 100, 0, [(10, M), (30000, N), (10, M)], 7.0, "."),
        (2, 0, 0, [(1, M)], 1.0, "."),
    ]
    _check(ctx, mk(recs), True)
    _check(ctx, mk(recs), False)
    # empty and all-unmapped inputs
    e = mk([])
    got = ctx.coverage(e)
    assert got["n_intervals"] == 0 and got["n_junctions"] == 0
    got = ctx.coverage(mk([(0, 5, 4, [(10, M)], 1.0, ".")]))
    assert got["n_intervals"] == 0


def test_fractional_yc_ordered_path(ctx):
    """--store-frac style YC values: per-base sums must follow record order exactly."""
    from tiebrush_amd import synth
    from oracle import oracle_ffi as orc
    tile = synth.make_tile(2, 30000, "c5", n_loci=300)
    groups = orc.collapse(tile, keep_secondary=True, store_frac=True)
    cin = synth.collapsed_to_cov_input(tile, groups)
    assert np.any(cin.yc != np.floor(cin.yc))
    _check(ctx, cin, True)


def test_fatal_op_is_reported(ctx):
    from tiebrush_amd import soa, api
    cin = soa.CovInput(np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.uint16), np.array([0, 1], np.uint32),
                       np.array([(10 << 4) | 7], np.uint32), np.ones(1), np.array([46], np.uint8))
    with pytest.raises(api.TbkError) as ei:
        ctx.coverage(cin)
    assert ei.value.status == -7


@pytest.mark.parametrize("name", ["t1", "t2"])
def test_sample_track_golden(ctx, name, bam_loader):
    """tiecov -s: float32 running mean of YX per base in record order (ordered tile kernel)"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, soa
    b = bam_loader(os.path.join(GOLDEN, name, name + ".bam"))
    cin = soa.cov_input_from_bam(b)
    want = orc.coverage(cin, want_cov=False, want_junc=False, num_samples=10)
    for dev in (False, True):
        got = api.to_numpy(ctx.sample(api.to_device(cin, "cuda:0") if dev else cin, 10,
                                      cap_intervals=want["n_sample"] + 1000))
        assert got["n_sample"] == want["n_sample"]
        for k in ("s_tid", "s_start", "s_end", "s_count", "s_heat"):
            assert np.array_equal(got[k], want[k]), k


def test_sample_track_synthetic_deep(ctx):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth
    tile = synth.make_tile(6, 30000, "c2", n_loci=200)
    groups = orc.collapse(tile)
    cin = synth.collapsed_to_cov_input(tile, groups)
    want = orc.coverage(cin, want_cov=False, want_junc=False, num_samples=6)
    got = api.to_numpy(ctx.sample(cin, 6))
    assert got["n_sample"] == want["n_sample"]
    for k in ("s_tid", "s_start", "s_end", "s_count", "s_heat"):
        assert np.array_equal(got[k], want[k]), k


def test_deep_pileup_needs_64bit_accumulators(ctx):
    """2.4M records of YC 1000 on one spot: the depth (2.4e9) is beyond int32, and every thread of the prep kernel sees
    about nine records — the |YC| total that picks the accumulator width must count all of them."""
    from tiebrush_amd import soa
    n = 2_400_000
    M = 0
    pos = np.full(n, 1000, np.int32)
    pos[n // 2:] = 1010
    cin = soa.CovInput(tid=np.zeros(n, np.int32), pos=pos, flag=np.zeros(n, np.uint16),
                       cig_off=np.arange(n + 1, dtype=np.uint32), cig=np.full(n, (40 << 4) | M, np.uint32),
                       yc=np.full(n, 1000.0), strand=np.full(n, ord("."), np.uint8), yx=np.ones(n, np.int64))
    got = _check(ctx, cin, True)
    assert float(got["iv_val"].max()) == 1000.0 * n


@pytest.mark.parametrize("frac", [False, True])
def test_junctions_only_and_intervals_only(ctx, frac):
    """tiecov -j without -c and -c without -j: each branch alone (the junction branch then runs inline, not on the side
    context), integral and fractional YC."""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth
    tile = synth.make_tile(2, 60000, "c2", n_loci=300)
    g = orc.collapse(tile)
    cin = synth.collapsed_to_cov_input(tile, g)
    if frac:
        cin.yc = cin.yc + 0.25
    want = orc.coverage(cin)
    dc = api.to_device(cin, "cuda:0")
    j = api.to_numpy(ctx.coverage(dc, want_cov=False))
    for k in ("j_tid", "j_start", "j_end", "j_strand", "j_val"):
        assert np.array_equal(j[k], want[k]), k
    c = api.to_numpy(ctx.coverage(dc, want_junc=False))
    for k in ("iv_tid", "iv_start", "iv_end", "iv_val"):
        assert np.array_equal(c[k], want[k]), k
