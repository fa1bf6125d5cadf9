"""Multi-rank path with the real HIP compute: R virtual ranks on the one GPU of the box (loopback driver), checked
against the flat oracle result, in both protocols: "partials" (local collapse -> tbk_partial_* -> the owner's collapse of the
partials, window path in its PART form for tiles >= 64 Ki partials or under TBK_PATH=window) and "shuffle" (tbk_shard_*).  Host
tiles take the numpy restatement of the device steps around the HIP collapse; device-resident tiles the kernels end to end."""
import numpy as np
import pytest

from helpers import tbk_debug

from dist_helpers import split_tile, check_against_flat, STRAT

pytestmark = pytest.mark.gpu


class GpuCompute:
    def __init__(self):
        from tiebrush_amd import api
        self.ctx = api.Context(0)
        self.api = api

    def collapse(self, tile, **kw):
        return self.api.to_numpy(self.ctx.collapse(tile, **kw))

    def coverage(self, cin):
        return self.api.to_numpy(self.ctx.coverage(cin))


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
@pytest.mark.parametrize("world,profile,strategy,kw", [
    (2, "c2", "cigar", {}),
    (4, "c3", "clip", {}),
    (8, "c5", "exon", dict(max_nh=5, min_qual=1)),
])
def test_loopback_gpu_equals_flat_oracle(world, profile, strategy, kw, mode):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, synth
    tile = synth.make_tile(world * 2, 20000, profile, n_loci=800)
    flat = orc.collapse(tile, strategy=STRAT[strategy], **kw)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    res = dist.run_loopback(GpuCompute(), tiles, first, strategy=strategy, want_coverage=True, mode=mode, **kw)
    check_against_flat(res, tile, flat, flat_cov)


def test_loopback_gpu_golden(bam_loader):
    from helpers import sample_paths
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, soa
    bams = [bam_loader(p) for p in sample_paths("t1")]
    tile = soa.tile_from_bams(bams)
    flat = orc.collapse(tile)
    tiles, first = split_tile(tile, 4)
    res = dist.run_loopback(GpuCompute(), tiles, first)
    check_against_flat(res, tile, flat)


class DeviceCompute:
    """tiles resident in HBM, results stay torch tensors; tiecov through the device chain"""

    def __init__(self):
        from tiebrush_amd import api
        self.ctx = api.Context(0)

    def collapse(self, tile, **kw):
        return self.ctx.collapse(tile, **kw)

    def groups_to_cov_in(self, fin):
        return self.ctx.groups_to_cov_in(fin)

    def __getattr__(self, name):          # tbk_shard_* / tbk_partial_*
        if name.startswith("shard_") or name.startswith("partial_"):
            return getattr(self.ctx, name)
        raise AttributeError(name)

    def finish_yd(self):
        self.ctx.finish_yd()

    def coverage(self, view):
        from tiebrush_amd import api
        return api.to_numpy(self.ctx.coverage(view))


@pytest.mark.parametrize("mode,path", [("partials", None), ("partials", "general"), ("partials", "general-window"), ("shuffle", None)])
@pytest.mark.parametrize("world,nfiles,profile,strategy,kw", [
    (4, 8, "c2", "cigar", {}),
    (3, 7, "c3", "clip", {}),
    (8, 16, "c5", "exon", dict(max_nh=5, min_qual=1)),
    (2, 2, "c2", "cigar", dict(keep_secondary=True)),
    (1, 3, "c2", "cigar", {}),
    (8, 256, "c2", "cigar", {}),      # BASELINE.json configs[3]'s shape: 256 files, 32 per rank, 8 ranks
    (8, 1024, "c5", "exon", dict(max_nh=5, min_qual=1)),   # configs[4]'s shape: 1024 files over 8 ranks
])
def test_loopback_device_resident_equals_flat_oracle(world, nfiles, profile, strategy, kw, mode, path, monkeypatch):
    """path None: the owner reduces with tbk_partial_reduce (merge of the runs); "general": with tbk_partial_unpack +
    tbk_collapse_tile (TBK_PARTIAL_REDUCE=0; what a hashed-key collision or a pile-up of partials falls back to), sort path;
    "general-window": the same under TBK_PATH=window — the PART form of wg_hash_k / wg_hash2_k / wg_sort_k (wgroup.hip)"""
    if path:
        monkeypatch.setenv("TBK_PARTIAL_REDUCE", "0")
        if path.endswith("window"):
            tbk_debug(monkeypatch, path="window")
    import torch
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist, synth
    tile = synth.make_tile(nfiles, 20000 if nfiles <= 16 else (4000 if nfiles <= 256 else 1000), profile, n_loci=800)
    flat = orc.collapse(tile, strategy=STRAT[strategy], **kw)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]
    comp = DeviceCompute()
    # the device chain view lives in context memory until the next call: run coverage inside each rank's turn
    res = dist.run_loopback(comp, dtiles, first, strategy=strategy, want_coverage=True, device_chain=True, mode=mode, **kw)
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat, flat_cov)


@pytest.mark.parametrize("strategy,kw", [("cigar", {}), ("exon", dict(max_nh=5, min_qual=1))])
@pytest.mark.parametrize("general", [False, True])
def test_loopback_8_ranks_32x200k_per_rank(strategy, kw, general, monkeypatch):
    """8 virtual ranks x 32 files x 200 k reads (51 M records, generated on the GPU): every local collapse takes the raw window
    path, every owner reduces ~ 0.5-1 M partials — by tbk_partial_reduce, or (general) by the PART window form of
    tbk_collapse_tile — exact against the flat oracle run"""
    if general:
        monkeypatch.setenv("TBK_PARTIAL_REDUCE", "0")
    import torch
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, synth, synth_dev
    world, fpr, reads = 8, 32, 200_000
    profile = "c2" if strategy == "cigar" else "c5"
    tx = synth.make_transcriptome()
    dtiles = [synth_dev.make_tile_device(fpr, reads, profile, device="cuda:0", first_file=r * fpr, tx=tx) for r in range(world)]
    first = [r * fpr for r in range(world)]
    comp = DeviceCompute()
    res = dist.run_loopback(comp, dtiles, first, strategy=strategy, want_coverage=True, device_chain=True, **kw)
    assert max(r.n_partials_received for r in res) >= 65536
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    host = [synth_dev.tile_to_host(t) for t in dtiles]
    del dtiles
    torch.cuda.empty_cache()
    from tiebrush_amd.soa import SoATile
    fo = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(h.file_off.astype(np.int64)) for h in host]))]).astype(np.uint32)
    co = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(h.cig_off.astype(np.int64)) for h in host]))]).astype(np.uint32)
    cat = lambda name: np.concatenate([getattr(h, name) for h in host])
    tile = SoATile(n_files=world * fpr, file_off=fo, tbmerged=np.zeros(world * fpr, np.uint8), tid=cat("tid"), pos=cat("pos"), flag=cat("flag"),
                   mapq=cat("mapq"), strand=cat("strand"), nh=cat("nh"), cig_off=co, cig=cat("cig"))
    del host
    flat = orc.collapse(tile, strategy=STRAT[strategy], **kw)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    check_against_flat(res, tile, flat, flat_cov)


def test_part_form_refuses_what_it_cannot_hold(monkeypatch):
    """a partial tile with a fractional carried YC under TBK_PATH=window: the PART kernels raise TBK_DERR_FRACTIONAL and the tile
    takes the sort path's ordered sums — the result is the oracle's"""
    import os
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, synth
    tile = synth.make_tile(3, 30000, "c3", n_loci=300)
    n = tile.n_records
    rng = np.random.default_rng(9)
    tile.tbmerged = np.ones(3, np.uint8)
    tile.yc_in = (rng.integers(1, 9, n) / 4.0).astype(np.float32).astype(np.float64)
    tile.yx_in = rng.integers(1, 4, n).astype(np.int64)
    tile.yd_in = rng.integers(0, 50, n).astype(np.int64)
    tile.prio_hi = rng.integers(0, 1000, n).astype(np.uint64)
    tile.prio_lo = np.arange(n).astype(np.uint64)       # (as in a real partial tile: the low word grows with the tile index)
    from dist_helpers import OracleCompute
    want = OracleCompute().collapse(tile, strategy="clip")
    ctx = api.Context(0)
    tbk_debug(monkeypatch, path="window")
    got = api.to_numpy(ctx.collapse(api.to_device(tile, "cuda:0"), strategy="clip"))
    tile.yc_in = np.rint(tile.yc_in * 4.0)                     # integral: the PART form holds it
    want2 = OracleCompute().collapse(tile, strategy="clip")
    got2 = api.to_numpy(ctx.collapse(api.to_device(tile, "cuda:0"), strategy="clip"))
    tbk_debug(monkeypatch, path=None)
    for g, w in ((got, want), (got2, want2)):
        assert g["n_groups"] == w["n_groups"]
        for k in ("rep", "yc", "yx", "yd", "g_start", "g_end"):
            assert np.array_equal(np.asarray(g[k]), np.asarray(w[k])), k
    ctx.close()


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
@pytest.mark.parametrize("world", [2, 4])
def test_loopback_device_resident_tbmerged_inputs(world, bam_loader, mode):
    """carried YC / YX / YD of TieBrush-merged inputs through the device shuffle (third all-to-all + gathered file flags)"""
    import os
    import torch
    from helpers import GOLDEN, sample_paths
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist, soa
    bams = [bam_loader(os.path.join(GOLDEN, "t1", "t1.bam"))] + [bam_loader(p) for p in sample_paths("t2")[:5]] + \
           [bam_loader(os.path.join(GOLDEN, "t2", "t2.bam"))]
    tile = soa.tile_from_bams(bams)
    flat = orc.collapse(tile)
    flat_cov = orc.coverage(__import__("tiebrush_amd.synth", fromlist=["x"]).collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]
    res = dist.run_loopback(DeviceCompute(), dtiles, first, want_coverage=True, device_chain=True, mode=mode)
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat, flat_cov)


@pytest.mark.parametrize("want_key", [True, False])
def test_partial_kernels_match_host_restatement(want_key):
    """tbk_partial_keys / _pack / _unpack against the numpy restatement dist.py uses for host tiles (with and without
    tbk_groups_out.g_key, which spares the kernels their gathers through the representative)"""
    import torch
    from tiebrush_amd import api, dist, synth
    tile = synth.make_tile(5, 20000, "c5", n_loci=400)
    ctx = api.Context(0)
    dt = api.to_device(tile, "cuda:0")
    kw = dict(strategy="exon", max_nh=5, min_qual=1)
    fin = ctx.collapse(dt, want_coords=True, want_effend=True, want_key=want_key, **kw)
    hfin = api.to_numpy(fin)
    key, emax, bad = ctx.partial_keys(dt, fin)
    hk, hm, hb = dist._partial_keys_np(tile, hfin)
    assert bad == hb == 0 and np.array_equal(key.cpu().numpy(), hk) and np.array_equal(emax.cpu().numpy(), hm)
    ng = fin["n_groups"]
    cuts_h = np.array([hk[ng // 3], hk[(2 * ng) // 3] + 1, dist.KEY_INF], np.int64)
    cuts = torch.from_numpy(cuts_h).cuda()
    rows, cigw, tab = ctx.partial_pack(dt, fin, key, cuts, 4, 100, **kw)
    hr, hc, ht = dist._partial_pack_np(tile, hfin, hk, cuts_h, 4, 100)
    gr = rows.cpu().numpy()
    assert np.array_equal(tab.cpu().numpy(), ht) and np.array_equal(gr[:, :10], hr[:, :10]) and not gr[:, 11].any()
    assert (gr[:, 10] != 0).all()                      # the key word: an exact code (bit 31) or a 31-bit hash, never absent
    assert np.array_equal(cigw.cpu().numpy()[:len(hc)].view(np.uint32), hc)
    A = ctx.partial_unpack(rows)
    hA = dist._partial_unpack_np(hr)
    for k, v in hA.items():
        assert np.array_equal(A[k].cpu().numpy().view(v.dtype), v), k
    ctx.close()


@pytest.mark.parametrize("profile,strategy,kw", [("c2", "cigar", {}), ("c3", "clip", {}), ("c5", "exon", dict(max_nh=5, min_qual=1)),
                                                 ("c5", "cigar", dict(keep_secondary=True, keep_supplementary=True))])
@pytest.mark.parametrize("runs", [1, 2, 5])
def test_partial_reduce_equals_general_path(profile, strategy, kw, runs):
    """tbk_partial_reduce against tbk_partial_unpack + tbk_collapse_tile on the same received rows: every output array, the
    representative included"""
    import torch
    from tiebrush_amd import api, synth
    from tiebrush_amd._lib import TbkError
    from tiebrush_amd.soa import SoATile
    tile = synth.make_tile(runs * 2, 20000, profile, n_loci=150)      # deep: many groups per position, ties on (strand, end)
    tiles, first = split_tile(tile, runs)
    ctx = api.Context(0)

    local = []
    for r in range(runs):
        dt = api.to_device(tiles[r], "cuda:0")
        fin = ctx.collapse(dt, strategy=strategy, want_coords=True, want_effend=True, **kw)
        local.append((dt, fin, ctx.partial_keys(dt, fin)[0]))

    def received():
        rows_all, cig_all, cnt = [], [], []
        for r, (dt, fin, key) in enumerate(local):
            rows, cigw, tab = ctx.partial_pack(dt, fin, key, None, 1, first[r], strategy=strategy, **kw)
            th = tab.cpu().numpy()
            rows_all.append(rows.clone())
            cig_all.append(cigw[:int(th[0, 2])].clone())
            cnt.append(int(th[0, 1]))
        return torch.cat(rows_all), torch.cat(cig_all), np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)

    rows, cig, ro = received()
    A = {k: v.clone() for k, v in ctx.partial_unpack(rows).items()}
    t2 = SoATile(n_files=runs, file_off=ro, tbmerged=np.ones(runs, np.uint8), tid=A["tid"], pos=A["pos"], flag=A["flag"], mapq=A["mapq"],
                 strand=A["strand"], nh=A["nh"], cig_off=A["cig_off"], cig=cig, yc_in=A["yc_in"], yx_in=A["yx_in"], yd_in=A["yd_in"],
                 prio_hi=A["prio_hi"], prio_lo=A["prio_lo"])
    want = api.to_numpy(ctx.collapse(t2, strategy=strategy, want_coords=True, keep_supplementary=True, keep_secondary=True))
    got = ctx.partial_reduce(rows, ro, cig, strategy=strategy)
    view = got.pop("view")
    got = api.to_numpy(got)
    assert got["n_groups"] == want["n_groups"] > 0
    for k in ("rep", "yc", "yx", "yd", "g_start", "g_end"):
        assert np.array_equal(np.asarray(got[k]).astype(np.int64), np.asarray(want[k]).astype(np.int64)), k
    # the view is what tbk_groups_to_cov_in builds from the general path's result: same coverage
    cov_a = api.to_numpy(ctx.coverage(view))
    fin2 = ctx.collapse(t2, strategy=strategy, want_coords=True, keep_supplementary=True, keep_secondary=True)
    cov_b = api.to_numpy(ctx.coverage(ctx.groups_to_cov_in(fin2)))
    for k in ("iv_tid", "iv_start", "iv_end", "iv_val", "j_tid", "j_start", "j_end", "j_strand", "j_val"):
        assert np.array_equal(cov_a[k], cov_b[k]), k
    ctx.close()


def test_partial_reduce_refuses_a_shared_hashed_key_word(monkeypatch):
    """two different three-exon alignments with equal (tid, start, strand, span) on two ranks: distinct groups; with the hash
    word masked away (TBK_DEBUG_HASH_MASK=0) they share a key word and tbk_partial_reduce must refuse (TBK_ECOLLISION) — never
    merge them — and the multi-rank driver then takes the general path, which reseeds"""
    import torch
    from helpers import tile_from_records
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist
    from tiebrush_amd._lib import TbkError
    M, N = 0, 3
    a = [(10, M), (20, N), (10, M), (30, N), (10, M)]
    b = [(10, M), (30, N), (10, M), (20, N), (10, M)]
    f0 = [(0, 100, 0, 60, "+", 1, a), (0, 100, 0, 60, "+", 1, a), (0, 300, 0, 60, "+", 1, [(50, M)])]
    f1 = [(0, 100, 0, 60, "+", 1, b), (0, 300, 0, 60, "+", 1, [(50, M)])]
    tile = tile_from_records([f0, f1])
    flat = orc.collapse(tile)
    assert flat["n_groups"] == 3
    tiles, first = split_tile(tile, 2)
    ctx = api.Context(0)
    for mask, refuse in ((None, False), ("0", True)):
        if mask is not None:
            tbk_debug(monkeypatch, hash_mask=mask)
        rows_all, cig_all, cnt = [], [], []
        for r in range(2):
            dt = api.to_device(tiles[r], "cuda:0")
            tbk_debug(monkeypatch, hash_mask=None)
            fin = ctx.collapse(dt, want_coords=True, want_effend=True)
            key = ctx.partial_keys(dt, fin)[0]
            if mask is not None:
                tbk_debug(monkeypatch, hash_mask=mask)
            rows, cigw, tab = ctx.partial_pack(dt, fin, key, None, 1, first[r])
            th = tab.cpu().numpy()
            rows_all.append(rows.clone())
            cig_all.append(cigw[:int(th[0, 2])].clone())
            cnt.append(int(th[0, 1]))
        rows, cig, ro = torch.cat(rows_all), torch.cat(cig_all), np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        if refuse:
            with pytest.raises(TbkError) as e:
                ctx.partial_reduce(rows, ro, cig)
            assert e.value.status == -8                  # TBK_ECOLLISION
        else:
            got = ctx.partial_reduce(rows, ro, cig)
            assert got["n_groups"] == 3 and got["yc"].cpu().numpy().tolist() == [2.0, 1.0, 2.0]
    # the driver: packed with the masked word, refused by the merge-reduce, reduced by the general path (whose own keys the mask
    # does not make exact either: it reseeds until the two alignments part, or gives up loudly) — here the mask is lifted for it
    tbk_debug(monkeypatch, hash_mask=None)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]
    res = dist.run_loopback(DeviceCompute(), dtiles, first, want_coverage=False, device_chain=True)
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat)
    ctx.close()


def test_shard_prepare_matches_host_restatement_and_rejects_unsorted():
    """tbk_shard_prepare against the numpy restatement dist.py uses for host tiles: keys, running maxima of the read
    ends, effective ends (computed before filtering) and filter verdicts; an unsorted file is refused."""
    import torch
    from tiebrush_amd import api, dist, synth
    from tiebrush_amd._lib import TbkError
    tile = synth.make_tile(3, 30000, "c5", n_loci=500)
    ctx = api.Context(0)
    kw = dict(max_nh=5, min_qual=1, keep_secondary=True)
    key, emax, effend, passm = ctx.shard_prepare(api.to_device(tile, "cuda:0"), **kw)
    hk, hm, he, hp = dist._prepare_np(tile, **kw)
    assert np.array_equal(key.cpu().numpy(), hk) and np.array_equal(emax.cpu().numpy(), hm)
    assert np.array_equal(effend.cpu().numpy(), he) and np.array_equal(passm.cpu().numpy() & 1, hp)
    assert 0 < int(hp.sum()) < tile.n_records
    bad = synth.make_tile(2, 1000, "c2", n_loci=50)
    bad.pos = bad.pos.copy()
    bad.pos[10], bad.pos[500] = bad.pos[500], bad.pos[10]
    with pytest.raises(TbkError) as e:
        ctx.shard_prepare(api.to_device(bad, "cuda:0"))
    assert e.value.status == -6                      # TBK_EUNSORTED
    with pytest.raises(ValueError):
        dist._prepare_np(bad)
    ctx.close()


def _two_proc_worker(rank, world, port, q, mode):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as td
    from dist_helpers import split_tile
    from test_gpu_dist import DeviceCompute
    from tiebrush_amd import api, dist, synth
    td.init_process_group("gloo", rank=rank, world_size=world)
    tile = synth.make_tile(6, 20000, "c3", n_loci=600)
    tiles, first = split_tile(tile, world)
    dt = api.to_device(tiles[rank], "cuda:0")
    r = dist.run_distributed(DeviceCompute(), dt, first[rank], want_coverage=True, device_chain=True, strategy="clip", mode=mode)
    for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
        v = getattr(r, f)
        setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    r.cov_input = None
    q.put((rank, r))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
def test_two_processes_device_path(mode):
    """Two real processes (torch.distributed, gloo staging the device tensors through the host) sharing the one GPU of
    the box, each with its own context and the HIP shuffle kernels: the multi-process protocol end to end.  (RCCL itself
    needs one GPU per rank and runs in the driver's scaling bench.)"""
    import os
    import torch.multiprocessing as mp
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_two_proc_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tile = synth.make_tile(6, 20000, "c3", n_loci=600)
    flat = orc.collapse(tile, strategy=STRAT["clip"])
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    check_against_flat([got[0], got[1]], tile, flat, flat_cov)


@pytest.mark.parametrize("mode", ["partials", "shuffle"])
@pytest.mark.parametrize("world", [1, 2, 4])
def test_loopback_device_real_bam_shapes(world, mode):
    """as tests/test_dist_cpu.py::test_loopback_real_bam_shapes, through the tbk_shard_* kernels (the keys stay monotone over
    unmapped and unplaced reads; an empty rank passes null arrays)"""
    import torch
    from helpers import paired_end_like_files, tile_from_records
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist, synth
    tile = tile_from_records(paired_end_like_files())
    flat = orc.collapse(tile)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]
    res = dist.run_loopback(DeviceCompute(), dtiles, first, want_coverage=True, device_chain=True, mode=mode)
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat, flat_cov)


def test_lists_protocol_settles_cuts_without_rounds():
    """the cut search on gathered bundle lists (tbk_partial_stage_*): no all-reduce round, five collectives and two read-backs a step,
    the result equal to the flat oracle — and to what the protocol's first form (cuts walked in rounds) delivers"""
    import torch
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist, synth
    world = 4
    tile = synth.make_tile(8, 30000, "c3", n_loci=900)
    flat = orc.collapse(tile, strategy=STRAT["clip"])
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]
    for mode in ("lists", "rounds"):
        sts = [dict() for _ in range(world)]
        res = dist.run_loopback(DeviceCompute(), dtiles, first, strategy="clip", want_coverage=True, device_chain=True, cut_search=mode,
                                per_rank=[dict(stats=sts[r]) for r in range(world)])
        for r in res:
            for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
                v = getattr(r, f)
                setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
        check_against_flat(res, tile, flat, flat_cov)
        if mode == "lists":
            assert all(st["cut_rounds"] == 0 and st["collectives"] == 6 and st["host_syncs"] == 2 for st in sts), sts   # (5 + the junction counts' gather)
        else:
            assert all(st["cut_rounds"] >= 1 for st in sts)


def test_lists_hand_a_long_chain_of_bundles_to_the_rounds():
    """two ranks whose bundles interlock like a staircase for more steps than a list describes: no candidate inside the lists' horizon
    is a clean cut, the cut stays unsettled (flag bit 1), and the protocol's first form walks it — same result as the flat oracle"""
    import torch
    from helpers import tile_from_records
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist
    M = 0
    # rank 0: reads [100 k, 100 k + 60), rank 1: [100 k + 50, 100 k + 110): every bundle of one rank bridges two of the other, 60 links long
    f0 = [(0, 10 + 5 * i, 0, 60, "+", 1, [(30, M)]) for i in range(40)] + [(0, 1000 + 100 * k, 0, 60, "+", 1, [(60, M)]) for k in range(60)] + \
         [(0, 20000 + 7 * i, 0, 60, "+", 1, [(30, M)]) for i in range(40)]
    f1 = [(0, 12 + 5 * i, 0, 60, "-", 1, [(30, M)]) for i in range(40)] + [(0, 1050 + 100 * k, 0, 60, "-", 1, [(60, M)]) for k in range(60)] + \
         [(0, 20001 + 7 * i, 0, 60, "-", 1, [(30, M)]) for i in range(40)]
    tile = tile_from_records([f0, f1])
    flat = orc.collapse(tile)
    tiles, first = split_tile(tile, 2)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]
    sts = [dict(), dict()]
    res = dist.run_loopback(DeviceCompute(), dtiles, first, device_chain=True, per_rank=[dict(stats=sts[r]) for r in range(2)])
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat)
    assert all(st["cut_rounds"] >= 1 for st in sts), sts          # the lists gave up, the rounds settled the cut
    assert res[0].n_groups > 0 and res[1].n_groups > 0


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("path", [None, "general"])
def test_loopback_full_strategy_with_md(world, path, bam_loader, monkeypatch):
    """-L across ranks: the golden t2 samples (MD tags on every record) sharded over virtual ranks, device-resident — the MD strings
    ride beside the partial rows, the owner's merge compares CIGAR then MD — against the flat oracle run with strategy FULL; with the
    owner's merge-reduce (tbk_partial_reduce_md) and with the general path (the PART collapse of a tile that carries the MD columns)"""
    import torch
    from helpers import sample_paths
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist, soa
    if path:
        monkeypatch.setenv("TBK_PARTIAL_REDUCE", "0")
    bams = [bam_loader(p, keep_md=True) for p in sample_paths("t2")[:6]]
    tile = soa.tile_from_bams(bams, with_md=True)
    # (the fixtures' reads are error-free: one MD string per CIGAR.  Give the records MD strings that differ inside a CIGAR group —
    # three variants and "no MD tag" —, so that the MD compare decides groups, their order inside a tie and the collisions of hashed keys)
    rng = np.random.default_rng(41)
    n = tile.n_records
    variants = [b"100", b"50A49", b"10^AC90", b""]
    pick = rng.integers(0, 5, n)
    mds = [variants[v] if v < 4 else None for v in pick]
    lens = np.array([0 if m is None else len(m) for m in mds], np.uint32)
    tile.md_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    tile.md = np.frombuffer(b"".join(m for m in mds if m), dtype=np.uint8).copy()
    tile.md_has = np.array([0 if m is None else 1 for m in mds], np.uint8)
    flat = orc.collapse(tile, strategy=STRAT["full"])
    assert flat["n_groups"] > orc.collapse(tile)["n_groups"]
    tiles, first = split_tile(tile, world)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]
    sts = [dict() for _ in range(world)]
    res = dist.run_loopback(DeviceCompute(), dtiles, first, strategy="full", device_chain=True, per_rank=[dict(stats=sts[r]) for r in range(world)])
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat)


def test_one_rank_on_the_general_path_posts_the_same_collectives(monkeypatch):
    """junction_gather=False (bench.py's carry scheme): a rank whose owner reduce is refused (a per-rank event: a shared hashed key word,
    a pile-up of partials) reduces through the general path — and must still post exactly the collectives its peers post.  Before the fix
    that path gathered the junction counts unconditionally: one rank in an all_gather its peers never enter (the loopback driver's
    "ranks diverged" assertion; a hang under torch.distributed)."""
    from oracle import oracle_ffi as orc
    from tiebrush_amd import api, dist, synth
    from tiebrush_amd._lib import TbkError
    tile = synth.make_tile(6, 8000, "c3", n_loci=300)
    flat = orc.collapse(tile, strategy=STRAT["clip"])
    tiles, first = split_tile(tile, 3)
    dtiles = [api.to_device(t, "cuda:0") for t in tiles]

    class OneRankRefused(DeviceCompute):
        calls = 0

        def __getattr__(self, name):
            if name == "partial_reduce":
                def refuse_second(*a, **k):
                    type(self).calls += 1
                    if type(self).calls == 2:       # rank 1's owner reduce
                        raise TbkError(-8, "forced")
                    return self.ctx.partial_reduce(*a, **k)
                return refuse_second
            return DeviceCompute.__getattr__(self, name)

    for jg in (False, True):
        OneRankRefused.calls = 0
        res = dist.run_loopback(OneRankRefused(), dtiles, first, want_coverage=True, device_chain=True, strategy="clip", junction_gather=jg)
        assert OneRankRefused.calls == 3
        assert sum(int(r.n_groups) for r in res) == flat["n_groups"]
        assert all(r.coverage is not None for r in res)
        if jg:
            nj = [int(r.coverage["n_junctions"]) for r in res]
            assert [r.junction_offset for r in res] == [0, nj[0], nj[0] + nj[1]]
