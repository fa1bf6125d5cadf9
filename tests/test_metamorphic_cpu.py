"""Metamorphic identities of the collapse on the CPU oracle (tests/metamorphic.py): what `-P`, `-E`, `-N`, `-Q` must do follows
from tiebrush.cpp:312-345 and :532-541,573 alone — the oracle's strategy and filter code is cross-checked against its own
default mode, which the reference's golden BAMs pin."""
import numpy as np
import pytest

import metamorphic as mm
from oracle import oracle_ffi as orc
from tiebrush_amd import synth

KEYS = ("rep", "yc", "yx", "yd", "g_start", "g_end")


def _same(a, b):
    assert a["n_groups"] == b["n_groups"] and a["n_passed"] == b["n_passed"]
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("seed", [0, 5])
def test_clip_equals_default_on_the_stripped_tile(seed):
    tile = synth.make_tile(n_files=6, reads_per_file=30000, profile="c3", n_loci=300, seed_base=0x71EB0000 + 1000 * seed)
    assert np.any((tile.cig & 0xF) == mm.S_OP)
    clip = orc.collapse(tile, strategy=2)
    _same(clip, orc.collapse(mm.strip_soft_clips(tile), strategy=0))
    assert clip["n_groups"] < orc.collapse(tile, strategy=0)["n_groups"]   # (the clips do split groups in the default mode)


@pytest.mark.parametrize("profile", ["c3", "c5"])
def test_exon_forms_the_default_groups_on_an_MN_only_tile(profile):
    tile = synth.make_tile(n_files=5, reads_per_file=30000, profile=profile, n_loci=300)
    t = mm.strip_soft_clips(mm.drop_indel_reads(tile)) if profile == "c3" else mm.drop_indel_reads(tile)
    assert mm.has_only_MN(t)
    mm.same_groups_any_tie_order(orc.collapse(t, strategy=3, keep_secondary=True, keep_supplementary=True),
                                 orc.collapse(t, strategy=0, keep_secondary=True, keep_supplementary=True))


@pytest.mark.parametrize("strategy", [0, 3])
def test_nh_and_mapq_filters_equal_default_with_ballast(strategy):
    tile = synth.make_tile(n_files=7, reads_per_file=30000, profile="c5", n_loci=300)
    t2, dropped = mm.ballast(tile, 5, 1)
    assert dropped > 1000
    a = orc.collapse(tile, strategy=strategy, max_nh=5, min_qual=1)
    _same(a, orc.collapse(t2, strategy=strategy))
    assert a["n_passed"] == int(np.count_nonzero((t2.flag & 0x900) == 0))
