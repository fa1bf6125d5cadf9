import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_CACHE = {}


def load_bam_cached(path, **kw):
    from tiebrush_amd import bamio
    key = (path, tuple(sorted(kw.items())))
    if key not in _CACHE:
        _CACHE[key] = bamio.read_bam(path, **kw)
    return _CACHE[key]


@pytest.fixture(scope="session")
def bam_loader():
    return load_bam_cached


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Everything native is built in-tree (and travels with the snapshot); build it when a fresh checkout lacks it."""
    need = [os.path.join(ROOT, "tiebrush_amd", "_build", n) for n in ("libtbk.so", "tiebrush", "tiecov", "tbh_tool")]
    need += [os.path.join(ROOT, "oracle", "_build", n) for n in ("libtb_oracle.so", "tb_cpu_e2e")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()
