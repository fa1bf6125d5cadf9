#!/usr/bin/env python3
"""End-to-end (BAM files -> BAM / bedgraph files) timing of the drop-in command lines on synthetic config-2 input.
Reported separately from bench.py's kernel-path number: here BGZF inflate, aux scans, PCIe copies,
tag updates and BGZF deflate are all inside the clock."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tiebrush_amd import synth
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    d = tempfile.mkdtemp(prefix="tbk_e2e_")
    t0 = time.time()
    tile = synth.make_tile(n_files, reads, "c2")
    paths = synth.write_bams(tile, os.path.join(d, "in"))
    t_gen = time.time() - t0
    binp = os.path.join(ROOT, "tiebrush_amd", "_build")
    out = os.path.join(d, "out.bam")
    def run(extra_env):
        best, rr = None, None
        for _ in range(3):
            t0 = time.time()
            rr = subprocess.run([os.path.join(binp, "tiebrush"), "-o", out] + paths, capture_output=True, text=True, check=True,
                                env=dict(os.environ, TBK_TIMING="1", **extra_env))
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        return best, rr

    host_only, _ = run({"TBK_DEVICE_DECODE": "0"})
    fast, _ = run({"TBK_BAM_LEVEL": "1"})
    best, r = run({})
    t0 = time.time()
    subprocess.run([os.path.join(binp, "tiecov"), "-c", os.path.join(d, "cov"), "-j", os.path.join(d, "junc"), out], check=True)
    t_cov = time.time() - t0
    insz = sum(os.path.getsize(p) for p in paths)
    print(json.dumps({"workload": "%d files x %d reads (config 2)" % (n_files, reads), "tiebrush_wall_s": round(best, 3),
                      "records_per_s_end_to_end": round(tile.n_records / best, 1),
                      "host_decode_wall_s": round(host_only, 3), "level1_output_wall_s": round(fast, 3),
                      "records_per_s_end_to_end_level1": round(tile.n_records / fast, 1), "tiecov_wall_s": round(t_cov, 3),
                      "input_bam_bytes": insz, "output_bam_bytes": os.path.getsize(out), "summary": r.stderr.strip().split("\n")[-1], "phases": [l for l in r.stderr.split("\n") if l.startswith("timing") or l.startswith("device")][-3:],
                      "generation_s": round(t_gen, 1), "host_cores": os.cpu_count()}))


if __name__ == "__main__":
    main()
