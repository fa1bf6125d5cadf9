"""oracle/_build/tb_cpu_e2e — the files -> files CPU baseline of bench.py (the repo's host codec around the oracle, the reference's
main loop restated: tiebrush.cpp:557-601) — reproduces the reference's golden BAMs through the SURVEY.md §4.4 normaliser, so the
end-to-end CPU figure in the bench line is the time of a run that produces the reference's output."""
import os
import subprocess

import pytest

from helpers import GOLDEN, sample_paths

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.environ.get("TBK_TEST_CPU_E2E") or os.path.join(ROOT, "oracle", "_build", "tb_cpu_e2e")   # (the override: the sanitizer builds of tools/san_check.sh)


def _run(args):
    return subprocess.run([TOOL] + args, check=True, capture_output=True, text=True)


@pytest.mark.parametrize("name,n_in", [("t1", 416922), ("t2", 242910)])
def test_cpu_e2e_samples_equal_golden(tmp_path, name, n_in):
    from tiebrush_amd import bamio
    out = str(tmp_path / "o.bam")
    r = _run(["-A", "-o", out] + sample_paths(name))
    o, g = bamio.read_bam(out, keep_aux=True), bamio.read_bam(os.path.join(GOLDEN, name, name + ".bam"))
    assert o.n == g.n
    for i in range(g.n):
        assert bamio.record_identity(o, i) == bamio.record_identity(g, i), i
        gyc = g.yc[i] if g.has_yc[i] else 1.0
        assert o.has_yc[i] and o.yc[i] == gyc and o.yx[i] == g.yx[i] and o.yd[i] == g.yd[i], i
    assert "%d input records written as %d" % (n_in, o.n) in r.stderr
    assert o.header.is_tiebrush() and len(o.header.co_samples()) == 10


def test_cpu_e2e_recollapse_equals_golden_t12(tmp_path):
    from tiebrush_amd import bamio
    out = str(tmp_path / "t12.bam")
    _run(["-A", "-o", out, os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2.bam")])
    o, g = bamio.read_bam(out), bamio.read_bam(os.path.join(GOLDEN, "t12.bam"))
    assert o.n == g.n == 9491
    for i in range(g.n):
        assert bamio.record_identity(o, i) == bamio.record_identity(g, i)
        assert o.yx[i] == g.yx[i] and o.yd[i] == g.yd[i]


def test_cpu_e2e_one_core_is_one_thread(tmp_path):
    """pinned to one core (how bench.py runs the single-threaded line) the tool still produces the same records"""
    from tiebrush_amd import bamio
    out = str(tmp_path / "o.bam")
    cpu = sorted(os.sched_getaffinity(0))[0]
    subprocess.run([TOOL, "-o", out] + sample_paths("t2")[:3], check=True, capture_output=True, preexec_fn=lambda: os.sched_setaffinity(0, {cpu}))
    out2 = str(tmp_path / "o2.bam")
    _run(["-o", out2] + sample_paths("t2")[:3])
    a, b = bamio.read_bam(out), bamio.read_bam(out2)
    assert a.n == b.n and all(bamio.record_identity(a, i) == bamio.record_identity(b, i) for i in range(a.n))
    assert list(a.yx) == list(b.yx) and list(a.yd) == list(b.yd)
