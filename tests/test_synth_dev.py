"""The torch-op workload generator (tiebrush_amd/synth_dev.py) on the CPU: well-formed tiles, the SURVEY.md §8d model,
and the coordinate-window sub-tile used as the bounded CPU-baseline sample."""
import numpy as np
import pytest


@pytest.mark.parametrize("profile", ["c2", "c3", "c5"])
def test_tiles_are_well_formed(profile):
    from tiebrush_amd import synth_dev
    t = synth_dev.tile_to_host(synth_dev.make_tile_device(3, 40_000, profile, device="cpu"))
    t.validate()
    for f in range(t.n_files):
        a, b = int(t.file_off[f]), int(t.file_off[f + 1])
        k = (t.tid[a:b].astype(np.int64) << 32) | t.pos[a:b]
        assert (np.diff(k) >= 0).all()                       # SO:coordinate per file
    op, ln = t.cig & 15, t.cig >> 4
    assert set(np.unique(op)) <= {0, 1, 2, 3, 4} and (ln > 0).all()
    q = np.add.reduceat(np.where(np.isin(op, [0, 1, 4]), ln, 0), t.cig_off[:-1].astype(np.int64))
    assert np.isin(q, [100, 98] if profile == "c5" else [100]).all()   # 100-bp reads (a 2D read gives its two bases to the deletion)
    assert int((t.cig_off[1:] - t.cig_off[:-1]).max()) < 256
    if profile == "c3":
        assert (op == 4).any()
    if profile == "c5":
        assert set(np.unique(t.nh)) == {1, 2, 5, 20} and (t.flag & 0x900).any() and (op == 1).any() and (op == 2).any()


def test_files_are_independent_streams_and_collapse_like_the_numpy_model():
    from oracle import oracle_ffi as orc
    from tiebrush_amd import synth, synth_dev
    t = synth_dev.tile_to_host(synth_dev.make_tile_device(4, 100_000, "c3", device="cpu"))
    g = orc.collapse(t, strategy=2)
    ref = orc.collapse(synth.make_tile(4, 100_000, "c3"), strategy=2)
    assert g["n_passed"] == ref["n_passed"] == 400_000
    assert abs(g["n_groups"] - ref["n_groups"]) < 0.02 * ref["n_groups"]   # same model, different random stream


def test_window_subtile_is_the_files_restricted_to_the_region():
    from tiebrush_amd import synth_dev
    d = synth_dev.make_tile_device(3, 50_000, "c2", device="cpu")
    full = synth_dev.tile_to_host(d)
    w = synth_dev.tile_to_host(d, window=(1, 1_000_000, 9_000_000))
    w.validate()
    keep = (full.tid == 1) & (full.pos >= 1_000_000) & (full.pos < 9_000_000)
    assert w.n_records == int(keep.sum()) > 0
    assert np.array_equal(w.pos, full.pos[keep]) and np.array_equal(w.flag, full.flag[keep])
    nc = (full.cig_off[1:] - full.cig_off[:-1])[keep]
    assert np.array_equal(w.cig_off[1:] - w.cig_off[:-1], nc)
    fo = np.searchsorted(np.flatnonzero(keep), full.file_off[1:].astype(np.int64))
    assert np.array_equal(w.file_off[1:], fo.astype(np.uint32))
