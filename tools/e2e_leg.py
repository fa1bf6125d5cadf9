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


def leg(files, reads, profile, flags, seq, runs, device_decode, desc, host_writer=False):
    """lay the inputs down, run the command line `runs` times, return the JSON object of the leg"""
    import torch

    from tiebrush_amd import synth, synth_dev
    d = tempfile.mkdtemp(prefix="tbk_e2e_", dir="/tmp")
    try:
        t0 = time.time()
        dev = "cuda:0" if torch.cuda.is_available() else "cpu"
        tile = synth_dev.tile_to_host(synth_dev.make_tile_device(files, reads, profile, device=dev))
        if dev != "cpu":
            torch.cuda.empty_cache()
        paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=seq)
        os.sync()          # (the inputs' dirty pages go to the disk now, not under the first timed run)
        t_gen = time.time() - t0
        n = tile.n_records
        del tile
        binp = os.path.join(ROOT, "tiebrush_amd", "_build")
        out = os.path.join(d, "out.bam")

        def run(extra_env, k, extra_flags=()):
            ts, rr = [], None
            for _ in range(k):
                t1 = time.time()
                rr = subprocess.run([os.path.join(binp, "tiebrush"), "-o", out] + list(extra_flags) + flags + paths, capture_output=True, text=True, check=True,
                                    env=dict(os.environ, TBK_TIMING="1", **extra_env))
                ts.append(time.time() - t1)
                os.sync()      # (the output's dirty pages leave now, not under the next run's reads)
            return sorted(ts), rr

        ts, r = run({}, max(1, runs))
        med = ts[len(ts) // 2]
        lines = r.stderr.strip().split("\n")
        summary = next((l for l in reversed(lines) if "input records written as" in l), lines[-1])   # (the tool's own summary line)
        in_bytes = sum(os.path.getsize(p) for p in paths)
        res = {"value": round(n / med, 1), "unit": "records/s", "workload": desc % (files, reads),
               "wall_s": round(med, 3), "wall_s_min": round(ts[0], 3), "wall_s_max": round(ts[-1], 3), "runs": len(ts),
               "input_bam_bytes": in_bytes, "input_bytes_per_record": round(in_bytes / n, 1),
               "output_bam_bytes": os.path.getsize(out), "summary": summary,
               "phases": [l for l in r.stderr.split("\n") if l.startswith("host path") or l.startswith("hybrid path") or l.startswith("writer closed") or l.startswith("released")
                          or l.startswith("timing ms") or l.startswith("tiles:") or l.startswith("device writer")][-6:],
               "generation_s": round(t_gen, 1)}
        if device_decode:
            th, _ = run({"TBK_DEVICE_DECODE": "1"}, 1)
            res["device_decode_wall_s"] = round(th[0], 3)
        if host_writer:     # the same run with the output tagged and deflated by the cores (--writer host): what the device writer replaces
            th, rh = run({}, 1, ["--writer", "host"])
            res["host_writer_wall_s"] = round(th[0], 3)
            res["host_writer_output_bam_bytes"] = os.path.getsize(out)
            res["host_writer_phases"] = [l for l in rh.stderr.split("\n") if l.startswith("hybrid path") or l.startswith("host path")][-1:]
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=32)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--profile", default="c2")
    ap.add_argument("--no-extra", action="store_true", help="only the first leg (bare records, default collapse)")
    a = ap.parse_args()
    res = leg(a.files, a.reads, a.profile, [], False, a.runs, True, "%d files x %d reads (config-2 read model, records without SEQ), default collapse")
    res["measured"] = ("in this run: tools/e2e_leg.py, a child of bench.py that ended before bench.py touched the GPU; median of the runs; process "
                       "start, BGZF both ways, tagging and PCIe inside the clock")
    if not a.no_extra:
        # the same command line on records that carry SEQ / QUAL and an aligner's tags (about 240 inflated bytes per record, like the
        # reference's fixtures: what BGZF and the tagging really move), and with config 3's options on config 3's read model
        k2 = max(1, a.runs - 1)
        res["seq"] = leg(a.files, a.reads, a.profile, [], True, k2, True,
                         "%d files x %d reads (config-2 read model) WITH 100-bp SEQ / QUAL and aligner tags, default collapse", host_writer=True)
        # ... and four times as much of it: long enough for the ~ 0.3 s the HIP runtime takes to come up to stop being a third of the run
        res["seq_long"] = leg(2 * a.files, 2 * a.reads, a.profile, [], True, max(3, k2), False,
                              "%d files x %d reads (config-2 read model) WITH 100-bp SEQ / QUAL and aligner tags, default collapse")
        res["c3_options"] = leg(2 * a.files, max(1, a.reads // 2), "c3", ["--clip"], False, k2, False,
                                "%d files x %d reads (config-3 read model: 10 %% soft-clipped, records without SEQ), --clip")
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
