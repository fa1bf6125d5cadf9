"""Multi-rank path with the real HIP compute: R virtual ranks on the one GPU of the box (loopback driver), checked
against the flat oracle result.  Host tiles take the numpy restatement of the shuffle around the HIP collapse (explicit
priorities); device-resident tiles take the tbk_shard_* kernels end to end."""
import numpy as np
import pytest

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


@pytest.mark.parametrize("world,profile,strategy,kw", [
    (2, "c2", "cigar", {}),
    (4, "c3", "clip", {}),
    (8, "c5", "exon", dict(max_nh=5, min_qual=1)),
])
def test_loopback_gpu_equals_flat_oracle(world, profile, strategy, kw):
    from oracle import oracle_ffi as orc
    from tiebrush_amd import dist, synth
    tile = synth.make_tile(world * 2, 20000, profile, n_loci=800)
    flat = orc.collapse(tile, strategy=STRAT[strategy], **kw)
    flat_cov = orc.coverage(synth.collapsed_to_cov_input(tile, flat))
    tiles, first = split_tile(tile, world)
    res = dist.run_loopback(GpuCompute(), tiles, first, strategy=strategy, want_coverage=True, **kw)
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

    def __getattr__(self, name):          # tbk_shard_prepare / _probe_* / _pack / _unpack
        if name.startswith("shard_"):
            return getattr(self.ctx, name)
        raise AttributeError(name)

    def finish_yd(self):
        self.ctx.finish_yd()

    def coverage(self, view):
        from tiebrush_amd import api
        return api.to_numpy(self.ctx.coverage(view))


@pytest.mark.parametrize("world,nfiles,profile,strategy,kw", [
    (4, 8, "c2", "cigar", {}),
    (3, 7, "c3", "clip", {}),
    (8, 16, "c5", "exon", dict(max_nh=5, min_qual=1)),
    (2, 2, "c2", "cigar", dict(keep_secondary=True)),
    (1, 3, "c2", "cigar", {}),
    (8, 256, "c2", "cigar", {}),      # BASELINE.json configs[3]'s shape: 256 files, 32 per rank, 8 ranks (every rank's tile has 256 runs)
    (8, 1024, "c5", "exon", dict(max_nh=5, min_qual=1)),   # configs[4]'s shape: 1024 files over 8 ranks
])
def test_loopback_device_resident_equals_flat_oracle(world, nfiles, profile, strategy, kw):
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
    res = dist.run_loopback(comp, dtiles, first, strategy=strategy, want_coverage=True, device_chain=True, **kw)
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat, flat_cov)


@pytest.mark.parametrize("world", [2, 4])
def test_loopback_device_resident_tbmerged_inputs(world, bam_loader):
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
    res = dist.run_loopback(DeviceCompute(), dtiles, first, want_coverage=True, device_chain=True)
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat, flat_cov)


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


def _two_proc_worker(rank, world, port, q):
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
    r = dist.run_distributed(DeviceCompute(), dt, first[rank], want_coverage=True, device_chain=True, strategy="clip")
    for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
        v = getattr(r, f)
        setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    r.cov_input = None
    q.put((rank, r))
    td.barrier()
    td.destroy_process_group()


def test_two_processes_device_path():
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
    procs = [ctx.Process(target=_two_proc_worker, args=(r, 2, port, q)) for r in range(2)]
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


@pytest.mark.parametrize("world", [1, 2, 4])
def test_loopback_device_real_bam_shapes(world):
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
    res = dist.run_loopback(DeviceCompute(), dtiles, first, want_coverage=True, device_chain=True)
    for r in res:
        for f in ("tid", "start", "end", "rep_fidx", "rep_idx", "yc", "yx", "yd"):
            v = getattr(r, f)
            setattr(r, f, v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    check_against_flat(res, tile, flat, flat_cov)
