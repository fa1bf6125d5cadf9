"""The window path of the collapse stage (wgroup.hip) against the oracle, forced onto small tiles (TBK_PATH=window) in both of
its forms: RAW (keys, filter and effective ends computed inside the window kernels from the input records) and compacted
(TBK_RAW=0: key pass + scan + compaction first).  Small adversarial tiles (one window), synthetic tiles of many windows with
every tier (hash table, second table, LDS sort), pile-ups with unsorted ends (the effective-end scan across waves, rows and
chunks), unmapped records inside and behind the files, inputs the raw form hands back to the general path."""
import numpy as np
import pytest

from helpers import tbk_debug

from test_gpu_fuzz import STRATS, _cmp, _rand_tile

pytestmark = pytest.mark.gpu

M, I, D, N, S = 0, 1, 2, 3, 4


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["1", "0", "dense"], ids=["raw", "compacted", "raw-dense-verify"])
def window_mode(request, monkeypatch):
    """raw: the records whose key word is hashed are listed per window for the verification pass; raw-dense-verify: they are
    marked in a per-record array instead (TBK_WG_DENSE_VERIFY, the form -L and the record -> group map always take)"""
    tbk_debug(monkeypatch, path="window")
    tbk_debug(monkeypatch, raw="1" if request.param == "dense" else request.param)
    if request.param == "dense":
        tbk_debug(monkeypatch, wg_dense_verify="1")
    return request.param


from helpers import tile_from_records as _tile  # noqa: E402


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_small_tiles(ctx, window_mode, seed):
    rng = np.random.default_rng(8100 + seed)
    for _ in range(10):
        tile = _rand_tile(rng, with_tb=False)
        for strat in STRATS:
            _cmp(ctx, tile, strategy=strat)
        _cmp(ctx, tile, strategy=str(rng.choice(STRATS)), keep_secondary=True, keep_supplementary=True,
             max_nh=int(rng.choice([1, 5, 2**31 - 1])), min_qual=int(rng.choice([-1, 1, 31])))


@pytest.mark.parametrize("profile,files,reads,kw", [
    ("c3", 8, 20000, dict(strategy="clip")),                      # deep: hash tier
    ("c2", 2, 60000, dict()),                                     # shallow: second table / LDS sort tier
    ("c5", 40, 3000, dict(strategy="exon", max_nh=5, min_qual=1)),  # filters + spliced reads + many files
    ("c3", 70, 1500, dict(strategy="cigar")),
    ("c2", 400, 300, dict()),                                     # several files per thread in the piece tables
])
def test_many_windows(ctx, window_mode, profile, files, reads, kw):
    from tiebrush_amd import synth
    tile = synth.make_tile(files, reads, profile, n_loci=400)
    _cmp(ctx, tile, **kw)


@pytest.mark.parametrize("rank", ["buckets", "merge"])
def test_ranking_by_buckets_and_by_merge_sort(ctx, window_mode, rank, monkeypatch):
    """the groups of a window are ranked through 256 buckets of their (reference, start) word (wg_bucket_rank) or, where a window
    spans several references or TBK_WG_RANK_MERGE asks for it, by the merge sort: both against the oracle on tiles whose windows
    cross references (three contigs, few loci), whose groups pile up on single bases (every group of a window in one bucket) and
    whose windows hold more than 512 groups (two groups per thread)"""
    from tiebrush_amd import synth
    if rank == "merge":
        tbk_debug(monkeypatch, wg_rank_merge="1")
    rng = np.random.default_rng(77)
    for files, reads, profile, kw in ((8, 20000, "c3", dict(strategy="clip")), (2, 60000, "c2", dict()), (40, 3000, "c5", dict(strategy="exon"))):
        _cmp(ctx, synth.make_tile(files, reads, profile, n_loci=60), **kw)
    # one base, many shapes: soft clips of every length on both sides under the default strategy -> hundreds of groups with one start
    recs = []
    for f in range(6):
        rows = []
        for i in range(700):
            a, b = int(rng.integers(0, 30)), int(rng.integers(0, 30))
            cig = ([(a, S)] if a else []) + [(100 - a - b, M)] + ([(b, S)] if b else [])
            rows.append((1, 5000, 0, 60, "+", 1, cig))
        for i in range(300):
            rows.append((1, 5001 + i, 16, 60, "-", 1, [(100, M)]))
        recs.append(rows)
    _cmp(ctx, _tile(recs))


@pytest.mark.parametrize("split", ["by_list", "radix"])
def test_yd_items_by_list_and_by_radix_split(ctx, window_mode, split, monkeypatch):
    """the YD items of the window path reach their lists without a sort (<= 64 files: bit-matrix ranks, yd_lcount_k /
    yd_lscatter_k) or through the stable radix split (more files; TBK_YD_RADIX forces it): both against the oracle — '.' strands
    feeding two lists, 64 files (every list in use), tiles of more than one 1024-group block, empty lists"""
    from tiebrush_amd import synth
    if split == "radix":
        tbk_debug(monkeypatch, yd_radix="1")
    for files, reads, profile, kw in ((64, 1500, "c3", dict(strategy="clip")), (3, 40000, "c2", dict()), (33, 2500, "c5", dict(strategy="exon"))):
        tile = synth.make_tile(files, reads, profile, n_loci=300)
        tile.strand = tile.strand.copy()
        tile.strand[::7] = ord(".")                      # unstranded reads among the stranded ones
        want = _cmp(ctx, tile, **kw)
        assert int(np.asarray(want["yd"]).max()) > 0


def test_pileup_effective_ends(ctx, window_mode):
    """One base, thousands of reads per file whose ends go up and down in file order: the representative is the member with the
    smallest (running maximum of the ends before it in its file, record index) — the scan's carry crosses lanes, waves,
    rows and chunks of the window kernel, and filtered / unmapped records take part in it exactly as the merge sees them."""
    rng = np.random.default_rng(5)
    files = []
    for f in range(3):
        recs = []
        for i in range(int(rng.integers(2500, 5200))):
            ln = int(rng.choice([30, 30, 40, 50, 50, 75, 90]))
            flag = int(rng.choice([0, 0, 0, 16, 0x100, 4]))          # secondary: filtered, still raises the running end
            mapq = int(rng.choice([60, 60, 0]))
            recs.append((1, 1000, flag, mapq, str(rng.choice(["+", "-"])), 1, [(ln, M)]))
        pre = [(0, int(p), 0, 60, "+", 1, [(25, M)]) for p in sorted(rng.integers(0, 500, 300))]
        post = [(1, int(p), 0, 60, "-", 1, [(10, M), (100, N), (15, M)]) for p in sorted(rng.integers(1001, 1400, 700))]
        tail = [(-1, -1, 4, 0, ".", -(2**31), [])] * 5              # unplaced reads close a sorted BAM
        files.append(pre + recs + post + tail)
    tile = _tile(files)
    _cmp(ctx, tile)
    _cmp(ctx, tile, min_qual=1)
    _cmp(ctx, tile, strategy="clip", keep_secondary=True)


def test_nothing_passes(ctx, window_mode):
    """every record filtered (the raw form only learns that after its window kernels have run), an empty file among the inputs"""
    from tiebrush_amd import synth
    tile = synth.make_tile(3, 4000, "c3", n_loci=30)
    _cmp(ctx, tile, min_qual=200)
    recs = [(0, 100, 0, 60, "+", 1, [(50, M)]), (0, 120, 0, 60, "+", 1, [(30, M)])]
    _cmp(ctx, _tile([recs, [], recs]))
    _cmp(ctx, _tile([[], recs]))


def test_unmapped_mates_inside_a_run(ctx, window_mode):
    """an unmapped read placed at its mate's position sits inside a run of equal starts: it neither splits the run nor
    contributes an end"""
    recs = [(0, 100, 0, 60, "+", 1, [(50, M)]), (0, 100, 4 | 8, 0, ".", 1, []), (0, 100, 0, 60, "+", 1, [(30, M)]),
            (0, 100, 0, 60, "+", 1, [(50, M)]), (0, 120, 0, 60, "+", 1, [(30, M)])]
    other = [(0, 100, 0, 60, "+", 1, [(30, M)]), (0, 100, 0, 60, "+", 1, [(50, M)])]
    _cmp(ctx, _tile([recs, other]))
    _cmp(ctx, _tile([other, recs]))


def test_inputs_the_raw_form_hands_back(ctx, window_mode):
    """an inversion among records that do not pass is not an error (the merge order is fixed on what passes and what came
    before it), an inversion that reaches a passing record is TBK_EUNSORTED — the raw form must not decide either by itself"""
    from tiebrush_amd import api
    ok = [(0, 100, 0, 60, "+", 1, [(50, M)]), (0, 300, 0x100, 60, "+", 1, [(50, M)]), (0, 200, 0x100, 60, "+", 1, [(50, M)]),
          (0, 400, 0, 60, "+", 1, [(50, M)])]
    _cmp(ctx, _tile([ok, ok]))
    bad = [(0, 100, 0, 60, "+", 1, [(50, M)]), (0, 300, 0, 60, "+", 1, [(50, M)]), (0, 200, 0, 60, "+", 1, [(50, M)])]
    with pytest.raises(api.TbkError) as e:
        ctx.collapse(_tile([ok, bad]))
    assert e.value.status == -6                                     # TBK_EUNSORTED


def test_exact_key_codes(ctx, window_mode):
    """key words that carry an exact code instead of a hash (single operation, M N M, two exons): alignments that differ only
    in the split of the first block / the gap, and shapes just beyond the code's field widths (hashed and verified)"""
    recs = []
    for a, g in [(10, 100), (11, 99), (10, 101), (1023, 50), (1024, 49), (10, (1 << 20) - 1), (10, 1 << 20), (9, (1 << 20) + 1)]:
        total = 3_000_000
        recs.append((0, 500, 0, 60, "+", 1, [(a, M), (g, N), (total - a - g, M)]))
        recs.append((0, 500, 0, 60, "+", 1, [(a, M), (g, N), (total - a - g, M)]))
    recs.append((0, 500, 0, 60, "+", 1, [(3_000_000, M)]))
    recs.append((0, 500, 0, 60, "+", 1, [(1_000_000, M), (1_000_000, D), (1_000_000, M)]))
    recs.append((0, 500, 0, 60, "+", 1, [(3, S), (10, M), (100, N), (2_999_890, M), (2, S)]))
    for strat in STRATS[0:1] + STRATS[2:]:
        _cmp(ctx, _tile([recs, list(reversed(recs))]), strategy=strat)


def test_debug_hooks_reach_a_live_context(monkeypatch):
    """TBK_DEBUG is read when a context is created; this binding forwards a later change through tbk_set_debug: a small tile takes the
    sort path, the window kernels under path=window, the sort path again once the key is gone — seen in the kernels that ran"""
    from tiebrush_amd import api, synth
    tile = synth.make_tile(3, 4000, "c2", n_loci=50)
    c = api.Context(0)
    c.set_profiling(True)

    def kernels():
        c.collapse(tile)
        return set(c.kernel_times())

    assert not any(k.startswith("wg_") for k in kernels())
    tbk_debug(monkeypatch, path="window")
    assert any(k.startswith("wg_hash") for k in kernels())
    tbk_debug(monkeypatch, path=None)
    assert not any(k.startswith("wg_") for k in kernels())
    c2 = api.Context(0)                              # a context created while the variable is set starts with it
    tbk_debug(monkeypatch, path="window")
    c3 = api.Context(0)
    c3.set_profiling(True)
    c3.collapse(tile)
    assert any(k.startswith("wg_hash") for k in c3.kernel_times())
    for x in (c, c2, c3):
        x.close()
