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
    # N > 1 defaults to BASELINE config 4, one fixed job split over the ranks (here 2 x 2 files of 30 k reads)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["ranks_seen"] == 2 and d["config"]["workload"].startswith("c4: one job of 4 synthetic sorted BAMs")
    assert d["unit"] == "records/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["records_per_gpu"] == 60000
    # whole-job value: both ranks' records over the slowest rank's time
    assert abs(d["value"] - 2 * 60000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic"} <= set(d["roofline"])
    # the same per-rank workload through the plain path, the exchange's cost and its volume ride in the line
    assert d["plain_ms_per_step"] > 0 and d["shuffle_ms_per_step"] >= 0 and d["dist_mode"] == "partials"
    assert 0 < d["partials_per_step"] <= 60000                    # one row per local group (this shallow shape has few duplicates)
    assert d["partials_per_step"] * 44 <= d["wire_bytes_per_step"] <= d["partials_per_step"] * (40 + 4 * 8)


def test_bench_default_line_contract_small():
    """the N = 1 line on a small shape: end_to_end and cpu_baseline (1 thread + tiewrap-style processes) are measured in the run"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--profile", "c3", "--files-per-gpu", "6",
           "--reads-per-file", "40000", "--e2e-files", "3", "--e2e-reads", "20000", "--e2e-runs", "1", "--cov-prof-reps", "10"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["records_per_gpu"] == 240000
    e = d["end_to_end"]
    assert "error" not in e, e
    assert e["value"] > 0 and e["runs"] == 1 and "60000 input records written as" in e["summary"]
    es, ec = d["end_to_end_seq"], d["end_to_end_c3_options"]          # records with SEQ / QUAL; config 3's options
    assert es["value"] > 0 and "60000 input records written as" in es["summary"] and es["input_bytes_per_record"] > 3 * e["input_bytes_per_record"]
    assert ec["value"] > 0 and "--clip" in ec["workload"] and "60000 input records written as" in ec["summary"]
    el = d["end_to_end_seq_long"]                                      # four times the records of the SEQ / QUAL leg
    assert el["value"] > 0 and "240000 input records written as" in el["summary"] and el["workload"].startswith("6 files x 40000 reads")
    assert es["host_writer_wall_s"] > 0 and es["host_writer_output_bam_bytes"] > 0   # the same leg with --writer host rides along
    c = d["cpu_baseline"]
    assert c["cores"] == 1 and c["value"] > 0 and c["value_O0"] > 0 and c["sample"].startswith("the whole tile")
    assert c["parallel"]["cores"] >= 2 and c["parallel"]["value"] > 0
    rc = d["roofline_coverage"]
    assert rc["launches_measured"] >= 10 and rc["launch_us_min"] <= rc["avg_launch_us"] <= rc["launch_us_max"]
    # SURVEY.md §8(d)'s two kernel-path figures at the top level, the link's own roofline beside the second
    assert d["value_resident"] == d["value"] and d["value_host_to_host"] == d["kernel_path_host_to_host"]["value"] > 0
    rl = d["roofline_link"]
    assert rl["bound"] == "pcie" and rl["peak"] > 0 and 0 < rl["frac"] <= 1.05 and rl["bytes_per_step"] > 0
    assert set(d["value_definitions"]) >= {"value", "value_host_to_host"}
    # the files -> files CPU path on the SEQ / QUAL leg's own files: one pinned core, and tiewrap-style processes
    ce = c["end_to_end"]
    assert "error" not in ce, ce
    assert ce["cores"] == 1 and ce["value"] > 0 and "tb_cpu_e2e:" in ce["phases"][0] and "end_to_end_seq" in ce["beside"]
    assert ce["parallel"]["value"] > 0 and ce["parallel"]["cores"] >= 2 and "tiewrap" in ce["parallel"]["mode"]
    assert "cpu_baseline" not in es                                     # (moved under cpu_baseline, not left in the leg)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE (how the driver calls it) starts the two ranks itself: ONE line, n_gpus 2, ranks_seen 2"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TBK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--files-per-gpu", "2",
           "--reads-per-file", "30000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["scaling"] == "strong"
    assert d["config"]["records_per_gpu"] == 60000
    assert abs(d["value"] - 2 * 60000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02


def test_bench_four_ranks_config4_shape():
    """`python bench.py --gpus 4` as the driver calls it (no WORLD_SIZE: it starts its ranks), four ranks on the one GPU through the gloo hook —
    as many processes as the box's guard allows beside the test runner (six on the card at once; the eight-rank rehearsal runs on the CPU:
    tests/test_dist_cpu.py::test_gloo_world8_config4_shape_equals_flat) — on config 4's per-rank shape (32 files per rank): ONE line,
    every rank seen, the whole-job value, the protocol's counts."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TBK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--files-per-gpu", "32",
           "--reads-per-file", "3000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["ranks_seen"] == 4 and d["scaling"] == "strong"
    assert d["config"]["workload"].startswith("c4: one job of 128 synthetic sorted BAMs") and d["config"]["records_per_gpu"] == 96000
    assert abs(d["value"] - 4 * 96000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    assert d["dist_mode"] == "partials" and d["cut_rounds_max"] == 0 and d["collectives_per_step"] <= 5
