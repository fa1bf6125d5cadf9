"""bench.py's multi-rank path under its launch contract (python -m torch.distributed.run, one process per rank): two ranks
share the box's one GPU through the TBK_BENCH_BACKEND=gloo hook (collectives staged through the host); the measured runs use
RCCL with one GPU per rank.  Checks the contract of the JSON line, not the speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_contract():
    env = dict(os.environ, TBK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29800 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--files-per-gpu", "2", "--reads-per-file", "30000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "records/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["records_per_gpu"] == 60000
    # whole-job value: both ranks' records over the slowest rank's time
    assert abs(d["value"] - 2 * 60000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
