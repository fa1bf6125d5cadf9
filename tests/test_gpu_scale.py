"""GPU parity at BASELINE.json sizes.
  * config 2 at full size (2 x 1M reads) and a config-3 shaped tile (64 files, --clip): bit-exact vs the oracle;
  * a 32 x 1M tile (and, with TBK_FULL_SCALE=1, config 3 at its full 64 x 5M) through size-independent properties:
    count conservation, output order, rec_group consistency, idempotence of re-collapsing the output, and for
    tiecov the checksum  sum((end-start)*value) == sum(YC * M bases)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tiebrush_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _exact(ctx, tile, **kw):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth
    okw = dict(kw)
    if "strategy" in okw:
        okw["strategy"] = {"cigar": 0, "full": 1, "clip": 2, "exon": 3}[okw["strategy"]]
    want = orc.collapse(tile, want_rec_group=True, **okw)
    dt = api.to_device(tile, "cuda:0")
    res = ctx.collapse(dt, want_rec_group=True, **kw)
    got = api.to_numpy(res)
    assert got["n_groups"] == want["n_groups"] and got["n_passed"] == want["n_passed"]
    for k in ("rep", "yc", "yx", "yd", "g_start", "g_end", "rec_group"):
        assert np.array_equal(got[k], want[k]), k
    cw = orc.coverage(synth.collapsed_to_cov_input(tile, want))
    cg = api.to_numpy(ctx.coverage(ctx.groups_to_cov_in(res)))
    for k in ("iv_tid", "iv_start", "iv_end", "iv_val", "j_tid", "j_start", "j_end", "j_strand", "j_val"):
        assert np.array_equal(cg[k], cw[k]), k
    assert cg["n_bases"] == cw["n_bases"] and cg["span_bases"] == cw["span_bases"]


def test_config2_full_size_exact(ctx):
    from tiebrush_amd import synth
    _exact(ctx, synth.make_tile(2, 1_000_000, "c2"))


def test_config3_shape_exact(ctx):
    from tiebrush_amd import synth
    _exact(ctx, synth.make_tile(64, 150_000, "c3"), strategy="clip")


def test_config5_shape_exact(ctx):
    from tiebrush_amd import synth
    _exact(ctx, synth.make_tile(128, 40_000, "c5"), strategy="exon", max_nh=5, min_qual=1)


@pytest.mark.parametrize("profile,kw", [("c2", {}), ("c3", {"strategy": "clip"})])
def test_long_buckets_exact(ctx, profile, kw):
    """A handful of loci: thousands of reads start on the same base (rRNA / mitochondrial style pile-ups), so the run
    sort meets (tid,start) buckets far longer than its local window and hands them to the block-level bucket sort."""
    from tiebrush_amd import synth
    tile = synth.make_tile(4, 150_000, profile, n_loci=3)
    key = (tile.tid.astype(np.int64) << 32) | tile.pos.astype(np.int64)
    assert np.unique(key, return_counts=True)[1].max() > 2048
    _exact(ctx, tile, **kw)


def test_amplicon_pileup_exact(ctx):
    """300k reads starting on ONE base (deep amplicon): a single (tid,start) bucket far beyond what one block should sort —
    the run sort hands the tile back to the radix sort; plus a second, ordinary locus so that both kinds of bucket meet."""
    from tiebrush_amd import soa
    rng = np.random.default_rng(5)
    M, S, I = 0, 4, 1
    files = []
    for f in range(3):
        n_hot, n_bg = 100_000, 20_000
        pos = np.concatenate([np.full(n_hot, 5000), np.sort(rng.integers(9000, 12000, n_bg))]).astype(np.int32)
        kind = rng.integers(0, 6, n_hot + n_bg)
        ln = rng.choice([70, 90, 100, 120], n_hot + n_bg)
        cigs = []
        for k, l in zip(kind, ln):
            if k == 0:
                cigs.append([(l << 4) | M])
            elif k == 1:
                cigs.append([(3 << 4) | S, ((l - 3) << 4) | M])
            elif k == 2:
                cigs.append([(40 << 4) | M, (2 << 4) | I, ((l - 42) << 4) | M])
            else:
                cigs.append([((l - 10 * int(k)) << 4) | M, (5 << 4) | S])
        files.append((pos, cigs))
    n = sum(len(p) for p, _ in files)
    fo = np.zeros(4, np.uint32)
    fo[1:] = np.cumsum([len(p) for p, _ in files])
    allc = [c for _, cs in files for c in cs]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum([len(c) for c in allc])
    tile = soa.SoATile(n_files=3, file_off=fo, tbmerged=np.zeros(3, np.uint8), tid=np.zeros(n, np.int32),
                       pos=np.concatenate([p for p, _ in files]), flag=np.zeros(n, np.uint16), mapq=np.full(n, 60, np.uint8),
                       strand=np.full(n, ord("."), np.uint8), nh=np.ones(n, np.int32), cig_off=off,
                       cig=np.array([x for c in allc for x in c], np.uint32))
    _exact(ctx, tile)
    _exact(ctx, tile, strategy="clip")


def _properties(ctx, tile, **kw):
    """`tile`: a numpy SoATile or one whose arrays already live on the GPU (synth_dev)."""
    import torch
    from tiebrush_amd import api, soa
    n = tile.n_records
    dt = tile if api._is_torch(tile.tid) else api.to_device(tile, "cuda:0")
    res = ctx.collapse(dt, want_rec_group=True, **kw)
    g = res["n_groups"]
    # passes_options (tiebrush.cpp:532-541) counted on the device, independently of the kernels under test
    fl = dt.flag.to(torch.int32) & 0xFFFF
    ok = (fl & 0x4) == 0
    if not kw.get("keep_supplementary"):
        ok &= (fl & 0x800) == 0
    if not kw.get("keep_secondary"):
        ok &= (fl & 0x100) == 0
    ok &= dt.mapq.to(torch.int32) >= kw.get("min_qual", -1)
    ok &= torch.where(dt.nh == -(2**31), torch.zeros_like(dt.nh), dt.nh) <= kw.get("max_nh", 2**31 - 1)
    n_pass = int(ok.sum())
    del fl
    assert res["n_passed"] == n_pass
    yc, yx, yd, rep = res["yc"], res["yx"], res["yd"], res["rep"].to(torch.int64) & 0xFFFFFFFF
    assert float(yc.sum()) == float(n_pass)                       # every passing record is counted exactly once
    assert int(yx.min()) >= 1 and int(yx.max()) <= tile.n_files and int(yd.min()) >= 0
    # output order: buckets by (tid,start); inside a bucket strand code then end
    tid = dt.tid[rep].to(torch.int64)
    code = torch.where(dt.strand[rep] == 43, 0, torch.where(dt.strand[rep] == 45, 1, 2)).to(torch.int64)
    key = ((tid + 1) << 33) | (res["g_start"].to(torch.int64) << 2) | code
    assert bool((key[1:] >= key[:-1]).all())
    same = key[1:] == key[:-1]
    assert bool((res["g_end"][1:][same] >= res["g_end"][:-1][same]).all())
    del key, same, code, tid
    # rec_group: every record belongs to one group, group sizes add up to YC, the representative is a member
    rg = res["rec_group"].to(torch.int64)
    assert torch.equal(rg >= 0, ok) and int(rg.max()) == g - 1    # a record has a group iff it passes the filters
    assert torch.equal(torch.bincount(rg[ok], minlength=g).to(torch.float64), yc)
    assert torch.equal(rg[rep], torch.arange(g, device=rg.device))
    del rg, ok
    # tiecov checksum on the collapsed records
    view = ctx.groups_to_cov_in(res)
    cov = ctx.coverage(view)
    ni = cov["n_intervals"]
    area = ((cov["iv_end"] - cov["iv_start"]).to(torch.float64) * cov["iv_val"]).sum()
    co = dt.cig_off.to(torch.int64) & 0xFFFFFFFF
    ops = dt.cig.to(torch.int64) & 0xFFFFFFFF
    mlen = torch.where((ops & 0xF) == 0, ops >> 4, torch.zeros_like(ops))
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=ops.device), torch.cumsum(mlen, 0)])
    del mlen
    mb = csum[co[rep + 1]] - csum[co[rep]]
    del csum
    assert float(area) == float((mb.to(torch.float64) * yc.to(torch.float32).to(torch.float64)).sum())
    assert cov["n_bases"] == int(mb.sum())
    iv = (cov["iv_tid"].to(torch.int64) << 32) | cov["iv_start"].to(torch.int64)
    assert bool((iv[1:] > iv[:-1]).all()) and bool((cov["iv_val"] != 0).all())
    del iv, cov, view, mb
    # idempotence: the collapsed output, fed back as one TieBrush-merged file, collapses to itself (built on the device)
    nc = co[rep + 1] - co[rep]
    off = torch.zeros(g + 1, dtype=torch.int64, device=nc.device)
    torch.cumsum(nc, 0, out=off[1:])
    idx = torch.repeat_interleave(co[rep] - off[:-1], nc) + torch.arange(int(off[-1]), device=nc.device)
    t2 = soa.SoATile(n_files=1, file_off=np.array([0, g], np.uint32), tbmerged=np.ones(1, np.uint8), tid=dt.tid[rep].contiguous(),
                     pos=dt.pos[rep].contiguous(), flag=dt.flag[rep].contiguous(), mapq=dt.mapq[rep].contiguous(),
                     strand=dt.strand[rep].contiguous(), nh=dt.nh[rep].contiguous(), cig_off=off.to(torch.int32), cig=dt.cig[idx].contiguous(),
                     yc_in=yc.to(torch.float32).to(torch.float64), yx_in=yx.clone(), yd_in=yd.to(torch.int64))
    del idx, ops, co
    want = {k: res[k].clone() for k in ("yc", "yx", "yd", "g_start", "g_end")}
    again = ctx.collapse(t2, **kw)
    assert again["n_groups"] == g
    assert torch.equal(again["rep"].to(torch.int64) & 0xFFFFFFFF, torch.arange(g, device=nc.device))
    for k in ("yc", "yx", "yd", "g_start", "g_end"):
        assert torch.equal(again[k], want[k]), k
    return g, ni


def test_properties_32x1M(ctx):
    from tiebrush_amd import synth
    tile = synth.make_tile(32, 1_000_000, "c3")
    g, ni = _properties(ctx, tile, strategy="clip")
    assert 0 < g < tile.n_records


def test_properties_config3_full_64x5M(ctx):
    """BASELINE.json configs[2] at its full size (320 M records, one tile), generated on the GPU (synth_dev)."""
    import torch
    from tiebrush_amd import synth_dev
    tile = synth_dev.make_tile_device(64, 5_000_000, "c3", device="cuda:0")
    g, ni = _properties(ctx, tile, strategy="clip")
    assert 0 < g < tile.n_records
    del tile
    torch.cuda.empty_cache()


@pytest.mark.parametrize("files,reads,profile,kw", [
    (32, 2_000_000, "c2", {}),                                               # configs[3]: what one of the 8 ranks holds
    (128, 1_000_000, "c5", dict(strategy="exon", max_nh=5, min_qual=1)),     # configs[4]: 1024 / 8 files, --exon -N 5 -Q 1
])
def test_properties_per_rank_shapes_config4_config5(ctx, files, reads, profile, kw):
    """BASELINE.json configs[3] / [4] at the full size one rank sees, generated on the GPU; the filters of config 5 drop records,
    so the expected counts come from a device-side evaluation of passes_options"""
    import torch
    from tiebrush_amd import synth_dev
    tile = synth_dev.make_tile_device(files, reads, profile, device="cuda:0")
    g, ni = _properties(ctx, tile, **kw)
    assert 0 < g < tile.n_records
    del tile
    torch.cuda.empty_cache()


def test_properties_config4_full_256x2M_one_gpu(ctx):
    """BASELINE.json configs[3] as ONE job on one GPU (512 M records, 256 input files: more than 64, so the window path lists
    (group, sample) incidences and the YD items take the radix split) — the N = 1 end of the strong-scaling workload
    (`bench.py --gpus 1 --profile c4 --scaling strong`)."""
    import torch
    from tiebrush_amd import synth_dev
    tile = synth_dev.make_tile_device(256, 2_000_000, "c2", device="cuda:0")
    g, ni = _properties(ctx, tile)
    assert 0 < g < tile.n_records
    del tile
    torch.cuda.empty_cache()


def test_properties_config5_1024_files_one_job(ctx):
    """BASELINE.json configs[4] as ONE job on one GPU: all 1024 input files in one tile (the window path's widest form: 1024 pieces per
    window, incidence lists, the radix-split YD items), `--exon -N 5 -Q 1`, at 1024 x 250 k = 256 M records — a quarter of the config's
    1 M reads per file: the full 1.02 G records are inside the raw window path's 2^30-record bound but their 2.9 G CIGAR words are beyond
    its 2^30-word bound (the eight ranks of the config hold 128 files x 1 M each: the per-rank test above)."""
    import torch
    from tiebrush_amd import synth_dev
    tile = synth_dev.make_tile_device(1024, 250_000, "c5", device="cuda:0")
    g, ni = _properties(ctx, tile, strategy="exon", max_nh=5, min_qual=1)
    assert 0 < g < tile.n_records
    del tile
    torch.cuda.empty_cache()


def test_device_generator_matches_host_model(ctx):
    """synth_dev on the GPU == synth_dev on the CPU (counter-based integer stream), and the tile is oracle-exact."""
    from tiebrush_amd import synth_dev
    a = synth_dev.tile_to_host(synth_dev.make_tile_device(5, 60_000, "c3", device="cuda:0"))
    b = synth_dev.tile_to_host(synth_dev.make_tile_device(5, 60_000, "c3", device="cpu"))
    for k in ("tid", "pos", "flag", "mapq", "strand", "nh", "cig_off", "cig"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    _exact(ctx, a, strategy="clip")
