"""bench.py's own rank launcher (`--gpus N` without WORLD_SIZE), the parts that need no GPU: a rank that fails ends the run with a
non-zero status, and --gpus must agree with WORLD_SIZE when a launcher has set one."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_gpus_must_agree_with_world_size():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-e2e"], env=_env(WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


def test_failed_rank_ends_the_run():
    """no GPU here: every rank fails when it selects its device; the parent reports it, stops the others and exits non-zero"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the ranks would run: covered by tests/test_gpu_bench_ranks.py")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-e2e",
                        "--no-cpu-baseline", "--files-per-gpu", "1", "--reads-per-file", "1000"], env=_env(TBK_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "stopping the other ranks" in r.stderr
    assert not [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert time.time() - t0 < 240
