#!/usr/bin/env python3
"""End-to-end leg of bench.py (run as its child, before bench.py touches the GPU): BAM files -> `tiebrush` -> BAM file.

Lays down --files synthetic coordinate-sorted BAMs of --reads reads (the config-2 read model; generated on the GPU by
tiebrush_amd/synth_dev.py, encoded by `tbh_tool mkbam`) under /tmp, runs the `tiebrush` command line --runs times as child
processes and prints ONE JSON object: records per second of the median run with process start, BGZF inflate, decode, PCIe,
collapse, tagging and BGZF deflate inside the clock (SURVEY.md §8d "end-to-end").  TBK_TIMING=1 makes the tool print its
phase times, which ride along."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=32)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--profile", default="c2")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import torch

    from tiebrush_amd import synth, synth_dev
    d = tempfile.mkdtemp(prefix="tbk_e2e_", dir="/tmp")
    try:
        t0 = time.time()
        dev = "cuda:0" if torch.cuda.is_available() else "cpu"
        tile = synth_dev.tile_to_host(synth_dev.make_tile_device(a.files, a.reads, a.profile, device=dev))
        if dev != "cpu":
            torch.cuda.empty_cache()
        paths = synth.write_bams_fast(tile, os.path.join(d, "in"))
        t_gen = time.time() - t0
        n = tile.n_records
        del tile
        binp = os.path.join(ROOT, "tiebrush_amd", "_build")
        out = os.path.join(d, "out.bam")

        def run(extra_env, runs):
            ts, rr = [], None
            for _ in range(runs):
                t1 = time.time()
                rr = subprocess.run([os.path.join(binp, "tiebrush"), "-o", out] + paths, capture_output=True, text=True, check=True,
                                    env=dict(os.environ, TBK_TIMING="1", **extra_env))
                ts.append(time.time() - t1)
            return sorted(ts), rr

        ts, r = run({}, max(1, a.runs))
        med = ts[len(ts) // 2]
        th, _ = run({"TBK_DEVICE_DECODE": "1"}, 1)
        lines = r.stderr.strip().split("\n")
        summary = next((l for l in reversed(lines) if "input records written as" in l), lines[-1])   # (the tool's own summary line)
        res = {"value": round(n / med, 1), "unit": "records/s", "workload": "%d files x %d reads (config-2 read model), default collapse" % (a.files, a.reads),
               "wall_s": round(med, 3), "wall_s_min": round(ts[0], 3), "wall_s_max": round(ts[-1], 3), "runs": len(ts),
               "device_decode_wall_s": round(th[0], 3), "input_bam_bytes": sum(os.path.getsize(p) for p in paths),
               "output_bam_bytes": os.path.getsize(out), "summary": summary,
               "phases": [l for l in r.stderr.split("\n") if l.startswith("host path") or l.startswith("writer closed") or l.startswith("released")][-3:],
               "generation_s": round(t_gen, 1),
               "measured": "in this run: tools/e2e_leg.py, a child of bench.py that ended before bench.py touched the GPU; median of the runs; process "
                           "start, BGZF both ways, tagging and PCIe inside the clock"}
        print(json.dumps(res), flush=True)
    finally:
        if not a.keep:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
