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
SETTLE_S = float(os.environ.get("TBK_E2E_SETTLE_S", "1.0"))   # pause before every timed run (outside the clock)
sys.path.insert(0, ROOT)


def cpu_budget():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_end_to_end(paths, flags, reads, d, one_thread_files=8):
    """The CPU path beside the leg, on the SAME files: oracle/_build/tb_cpu_e2e — the repo's host codec around the oracle, the
    reference's main loop restated (TInputFiles::next -> GSamReader::next -> addPData / flushPData -> GSamWriter::write; tiebrush.cpp:557-601) —
    (1) on one core, as the reference runs (a bounded sample: the first `one_thread_files` inputs), and (2) the reference's best case,
    tiewrap-style (tiewrap.py:96-126): the inputs in batches, one single-threaded process per batch side by side, then one more run over
    the batch outputs as TieBrush-merged inputs.  Process start, BGZF both ways, parsing and tagging inside the clock, like the leg's."""
    tool = os.path.join(ROOT, "oracle", "_build", "tb_cpu_e2e")
    if not os.path.exists(tool):
        return {"error": "oracle/_build/tb_cpu_e2e is not built"}
    cpus = sorted(os.sched_getaffinity(0))
    pin = lambda c: (lambda: os.sched_setaffinity(0, {c}))
    res = {}
    sub = paths[:max(1, min(one_thread_files, len(paths)))]
    out1 = os.path.join(d, "cpu1.bam")
    t1 = time.time()
    r = subprocess.run([tool, "-o", out1] + flags + sub, capture_output=True, text=True, check=True, preexec_fn=pin(cpus[0]))
    dt = time.time() - t1
    res.update({"value": round(len(sub) * reads / dt, 1), "unit": "records/s", "cores": 1, "kind": "port",
                "sample": "the first %d of the leg's %d input files (%d records), one process pinned to one core: %.1f s" % (len(sub), len(paths), len(sub) * reads, dt),
                "phases": [l for l in r.stderr.split("\n") if l.startswith("tb_cpu_e2e:")][-1:]})
    os.remove(out1)
    procs = min(16, cpu_budget(), len(paths), len(cpus))
    if procs >= 2:
        bsz = -(-len(paths) // procs)                    # tiewrap: -b ceil(k / cores)
        batches = [paths[i:i + bsz] for i in range(0, len(paths), bsz)]
        outs = [os.path.join(d, "cpu_b%d.bam" % b) for b in range(len(batches))]
        t1 = time.time()
        ps = [subprocess.Popen([tool, "-o", outs[b]] + flags + batches[b], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               preexec_fn=pin(cpus[b % len(cpus)])) for b in range(len(batches))]
        rcs = [p.wait() for p in ps]
        t_b = time.time() - t1
        if any(rcs):
            res["parallel"] = {"error": "a batch process failed: %r" % (rcs,)}
        else:
            outp = os.path.join(d, "cpu_par.bam")
            t1 = time.time()
            subprocess.run([tool, "-o", outp] + flags + outs, capture_output=True, text=True, check=True, preexec_fn=pin(cpus[0]))
            t_f = time.time() - t1
            res["parallel"] = {"value": round(len(paths) * reads / (t_b + t_f), 1), "unit": "records/s", "cores": len(batches), "kind": "port",
                               "mode": "tiewrap-style: %d batches of %d files, one single-threaded process each side by side (%.1f s), then one run over "
                                       "the %d batch outputs as TieBrush-merged inputs (%.1f s); tiewrap.py:96-126" % (len(batches), bsz, t_b, len(batches), t_f),
                               "sample": "all %d input files of the leg (%d records)" % (len(paths), len(paths) * reads)}
            os.remove(outp)
        for o in outs:
            if os.path.exists(o):
                os.remove(o)
    return res


def leg(files, reads, profile, flags, seq, runs, device_decode, desc, host_writer=False, cpu_base=False):
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
                # a NEW output file every run: over an existing one the tool pays for the old file twice — fopen("wb") drops its cached
                # pages (20 ms per 335 MB) and the close of a file rewritten after a truncate starts the write-back at once (30 ms)
                if os.path.exists(out):
                    os.remove(out)
                # ... and a run of its own: the driver takes a process's device memory down AFTER the process has gone, and a successor
                # that starts inside that second waits for it — its context comes up 0.1-0.2 s late, its first large device allocation
                # takes 120-290 ms instead of 0.6 (tools/stall_probe.py: with the pause none of ten runs, without it every second one)
                time.sleep(SETTLE_S)
                t1 = time.time()
                rr = subprocess.run([os.path.join(binp, "tiebrush"), "-o", out] + list(extra_flags) + flags + paths, capture_output=True, text=True,
                                    env=dict(os.environ, TBK_TIMING="1", **extra_env))
                if rr.returncode != 0:
                    raise RuntimeError("tiebrush exited with %d: %s" % (rr.returncode, rr.stderr[-2000:]))
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
                          or l.startswith("timing ms") or l.startswith("tiles:") or l.startswith("device writer") or l.startswith("bam_release") or l.startswith("hybrid decode starts") or l.startswith("collapse phases") or l.startswith("YD stage ms") or l.startswith("representatives")][-12:],
               "generation_s": round(t_gen, 1)}
        if device_decode:
            th, _ = run({"TBK_DEVICE_DECODE": "1"}, 1)
            res["device_decode_wall_s"] = round(th[0], 3)
        if host_writer:     # the same run with the output tagged and deflated by the cores (--writer host): what the device writer replaces
            th, rh = run({}, 1, ["--writer", "host"])
            res["host_writer_wall_s"] = round(th[0], 3)
            res["host_writer_output_bam_bytes"] = os.path.getsize(out)
            res["host_writer_phases"] = [l for l in rh.stderr.split("\n") if l.startswith("hybrid path") or l.startswith("host path")][-1:]
        if cpu_base:
            try:
                res["cpu_baseline"] = cpu_end_to_end(paths, flags, reads, d)
            except Exception as e:      # a report beside the leg, never the leg itself
                res["cpu_baseline"] = {"error": repr(e)}
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
    ap.add_argument("--cpu-baseline", action="store_true", help="time the CPU files -> files path (oracle/_build/tb_cpu_e2e) on the SEQ / QUAL leg's files")
    a = ap.parse_args()
    res = leg(a.files, a.reads, a.profile, [], False, a.runs, True, "%d files x %d reads (config-2 read model, records without SEQ), default collapse")
    res["measured"] = ("in this run: tools/e2e_leg.py, a child of bench.py that ended before bench.py touched the GPU; median of the runs; process "
                       "start, BGZF both ways, tagging and PCIe inside the clock")
    if not a.no_extra:
        # the same command line on records that carry SEQ / QUAL and an aligner's tags (about 240 inflated bytes per record, like the
        # reference's fixtures: what BGZF and the tagging really move), and with config 3's options on config 3's read model
        k2 = max(1, a.runs)      # (three runs a leg: the median of two is the slower one)
        res["seq"] = leg(a.files, a.reads, a.profile, [], True, k2, True,
                         "%d files x %d reads (config-2 read model) WITH 100-bp SEQ / QUAL and aligner tags, default collapse", host_writer=True,
                         cpu_base=a.cpu_baseline)
        # ... and four times as much of it: long enough for the ~ 0.3 s the HIP runtime takes to come up to stop being a third of the run
        res["seq_long"] = leg(2 * a.files, 2 * a.reads, a.profile, [], True, max(3, k2), False,
                              "%d files x %d reads (config-2 read model) WITH 100-bp SEQ / QUAL and aligner tags, default collapse")
        res["c3_options"] = leg(2 * a.files, max(1, a.reads // 2), "c3", ["--clip"], False, k2, False,
                                "%d files x %d reads (config-3 read model: 10 %% soft-clipped, records without SEQ), --clip")
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
