"""Multi-rank path on CPU: loopback of R virtual ranks and a real world_size-2 gloo run, both checked against the
flat single-tile result (per-tile compute served by the oracle; sharding/exchange/stitch is the product code).
Both protocols run: "partials" (collapse locally, exchange group partials — the default) and "shuffle" (records shuffled by
coordinate, then collapsed — the fallback for carried fractional YC)."""
import os
import sys

import numpy as np
import pytest

from dist_helpers import OracleCompute, split_tile, check_against_flat, STRAT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
@pytest.mark.parametrize("world,profile,strategy,kw", [
    (2, "c2", "cigar", {}),
    (3, "c3", "clip", {}),
    (4, "c5", "exon", dict(max_nh=5, min_qual=1)),
    (8, "c2", "cigar", {}),
])
def test_loopback_equals_flat(world, profile, strategy, kw, mode):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, synth
    tile = synth.make_tile(max(world, 4) * 2, 6000, profile, n_loci=300)
    flat = orc.collapse(tile, strategy=STRAT[strategy], **kw)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    res = dist.run_loopback(OracleCompute(), tiles, first, strategy=strategy, want_coverage=True, mode=mode, **kw)
    check_against_flat(res, tile, flat, flat_cov)
    assert sum(1 for r in res if r.n_groups > 0) >= min(world, 2)   # the work really is spread
    if mode == "partials":     # one row per LOCAL GROUP travels, not one per record
        from dist_helpers import STRAT as _S
        local_groups = sum(orc.collapse(t, strategy=_S[strategy], **kw)["n_groups"] for t in tiles)
        assert sum(r.n_partials_received for r in res) == local_groups < flat["n_passed"]


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
def test_loopback_golden_t2(bam_loader, mode):
    from helpers import sample_paths
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, soa
    bams = [bam_loader(p) for p in sample_paths("t2")]
    tile = soa.tile_from_bams(bams)
    flat = orc.collapse(tile)
    tiles, first = split_tile(tile, 5)
    res = dist.run_loopback(OracleCompute(), tiles, first, mode=mode)
    check_against_flat(res, tile, flat)


def test_partials_golden_chain_equals_flat_and_golden(bam_loader):
    """the fixture chain t1s*.bam -> t1.bam is a flat run of the ten sample files; the partials protocol over
    R ranks must give that same golden BAM (records, order, YC / YX / YD) — and, unlike a tiewrap-style hierarchical run, the
    same representative records."""
    from helpers import GOLDEN, sample_paths, compare_groups_to_golden_bam
    from tiebrush_amd import dist, soa
    bams = [bam_loader(p) for p in sample_paths("t1")]
    tile = soa.tile_from_bams(bams, with_names=False)
    gold = bam_loader(os.path.join(GOLDEN, "t1", "t1.bam"))
    tiles, first = split_tile(tile, 4)
    res = dist.run_loopback(OracleCompute(), tiles, first)
    fo = tile.file_off.astype(np.int64)
    cat = lambda name: np.concatenate([np.asarray(getattr(r, name)) for r in res])
    got = dict(n_groups=sum(r.n_groups for r in res), rep=(fo[cat("rep_fidx").astype(np.int64)] + cat("rep_idx").astype(np.int64)),
               yc=cat("yc"), yx=cat("yx"), yd=cat("yd"))
    bad = compare_groups_to_golden_bam(got, tile, bams, gold)
    # HEAD default (-A off) differs from the 0.0.6 goldens by YC + 1 on exactly the records of SURVEY.md §4.4
    assert bad == [1930, 2210]


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
@pytest.mark.parametrize("world", [2, 3])
def test_loopback_tbmerged_inputs(world, bam_loader, mode):
    """TieBrush-merged inputs (t1.bam, t2.bam: carried YC / YX / YD) mixed with plain sample files: the carried tags
    follow the shuffled rows, the file flags are gathered, and the result is the flat re-collapse (SURVEY §8e chain)."""
    from helpers import GOLDEN, sample_paths
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, soa
    bams = [bam_loader(os.path.join(GOLDEN, "t1", "t1.bam"))] + [bam_loader(p) for p in sample_paths("t2")[:3]] + \
           [bam_loader(os.path.join(GOLDEN, "t2", "t2.bam"))]
    tile = soa.tile_from_bams(bams)
    assert tile.tbmerged.tolist() == [1, 0, 0, 0, 1]
    flat = orc.collapse(tile)
    tiles, first = split_tile(tile, world)
    res = dist.run_loopback(OracleCompute(), tiles, first, mode=mode)
    check_against_flat(res, tile, flat)


def test_partials_fall_back_collectively_on_fractional_carried_yc():
    """a TieBrush-merged input written with --store-frac carries a fractional YC: a sum of sums would not keep the reference's
    order of additions, so every rank — also those whose own partials are integral — takes the record shuffle for the tile"""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, synth
    tile = synth.make_tile(4, 3000, "c2", n_loci=100)
    n = tile.n_records
    rng = np.random.default_rng(3)
    tile.tbmerged = np.array([0, 0, 0, 1], np.uint8)
    tile.yc_in = (rng.integers(1, 9, n) / 3.0).astype(np.float32).astype(np.float64)
    tile.yx_in = rng.integers(1, 4, n).astype(np.int64)
    tile.yd_in = rng.integers(0, 50, n).astype(np.int64)
    flat = orc.collapse(tile)
    assert np.any(flat["yc"] != np.rint(flat["yc"]))
    tiles, first = split_tile(tile, 2)
    res = dist.run_loopback(OracleCompute(), tiles, first)
    check_against_flat(res, tile, flat)
    assert sum(r.n_partials_received for r in res) == flat["n_passed"]      # records travelled, not partials


def test_refuses_order_dependent_flags():
    from tiebrush_amd import dist, synth
    tile = synth.make_tile(2, 100, "c2", n_loci=10)
    tiles, first = split_tile(tile, 2)
    with pytest.raises(ValueError):
        dist.run_loopback(OracleCompute(), tiles, first, store_frac=True)
    with pytest.raises(ValueError):                       # refused up front, before any collective has run
        dist.run_loopback(OracleCompute(), tiles, first, strategy="full")


def _gloo_tile():
    """4 files; file 1 poses as a TieBrush-merged input (carried YC / YX / YD), so the third all-to-all and the file-flag
    gather of the protocol run over the real process group too"""
    from tiebrush_amd import synth
    tile = synth.make_tile(4, 5000, "c2", n_loci=200)
    rng = np.random.default_rng(11)
    n = tile.n_records
    tile.tbmerged = np.array([0, 1, 0, 0], np.uint8)
    tile.yc_in = rng.integers(1, 40, n).astype(np.float64)
    tile.yx_in = rng.integers(1, 6, n).astype(np.int64)
    tile.yd_in = rng.integers(0, 90, n).astype(np.int64)
    return tile


def _c4_shape_tile():
    """BASELINE.json configs[3]'s shape at a size the CPU takes: 256 files (32 per rank at world 8), a few hundred reads each"""
    from tiebrush_amd import synth
    return synth.make_tile(256, 300, "c2", n_loci=120)


def _gloo_worker(rank, world, port, q, mode="partials", shape="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as td
    from dist_helpers import OracleCompute, split_tile
    from tiebrush_amd import dist
    from test_dist_cpu import _gloo_tile, _c4_shape_tile
    td.init_process_group("gloo", rank=rank, world_size=world)
    tile = _gloo_tile() if shape == "gloo" else _c4_shape_tile()
    tiles, first = split_tile(tile, world)
    r = dist.run_distributed(OracleCompute(), tiles[rank], first[rank], device="cpu", want_coverage=True, mode=mode)
    q.put((rank, r))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("mode,a2a_bytes", [("partials", None), ("partials", 4096), ("shuffle", None), ("shuffle", 4096)])
def test_gloo_world2_equals_flat(mode, a2a_bytes, monkeypatch):
    """a2a_bytes = 4096: every block of the exchange goes out in many bounded rounds (the path large tiles take)"""
    import torch.multiprocessing as mp
    if a2a_bytes:
        monkeypatch.setenv("TBK_A2A_MAX_BYTES", str(a2a_bytes))
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tile = _gloo_tile()
    flat = orc.collapse(tile)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    check_against_flat([got[0], got[1]], tile, flat, flat_cov)


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
@pytest.mark.parametrize("world", [1, 2, 4])
def test_loopback_real_bam_shapes(world, mode):
    """unmapped mates at their mate's position, unplaced reads at the end of a file, a rank whose files hold nothing, a CIGAR
    that ends in an intron right before a cut candidate: the multi-rank path takes them as the single-GPU path does"""
    from helpers import paired_end_like_files, tile_from_records
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, synth
    tile = tile_from_records(paired_end_like_files())
    flat = orc.collapse(tile)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    res = dist.run_loopback(OracleCompute(), tiles, first, want_coverage=True, mode=mode)
    check_against_flat(res, tile, flat, flat_cov)


def test_gloo_world8_config4_shape_equals_flat():
    """The first real run on eight GPUs is the driver's: this is its rehearsal on the CPU — EIGHT processes, torch.distributed (gloo), the
    partials protocol end to end on BASELINE config 4's shape (256 files, 32 per rank), every collective of the step at world 8,
    checked against the flat oracle run (records, representatives, YC / YX / YD, coverage and the junction numbering across ranks)."""
    import torch.multiprocessing as mp
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q, "partials", "c4")) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tile = _c4_shape_tile()
    flat = orc.collapse(tile)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    check_against_flat([got[r] for r in range(world)], tile, flat, flat_cov)
    assert sum(1 for r in range(world) if got[r].n_groups > 0) >= 6          # the key range really is spread over the ranks
