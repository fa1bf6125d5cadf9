#!/usr/bin/env python3
"""Where the seconds of a `tiebrush` run go that its own phase lines do not show: process start (fork / exec / loader) and what is left of
the exit after the output is closed.  Lays down the end_to_end_seq inputs, runs the command line with TBK_TIMING=1 TBK_EXIT_TIMING=1 (the
tool prints the wall-clock time of its _exit) and compares with the parent's clock around the child."""
import os
import re
import subprocess
import sys
import tempfile
import time
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from tiebrush_amd import synth, synth_dev
    files, reads = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    d = tempfile.mkdtemp(prefix="tbk_exit_", dir="/tmp")
    try:
        tile = synth_dev.tile_to_host(synth_dev.make_tile_device(files, reads, "c2", device="cuda:0"))
        torch.cuda.empty_cache()
        paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=True)
        del tile
        os.sync()
        exe = os.path.join(ROOT, "tiebrush_amd", "_build", "tiebrush")
        for extra in ({}, {"TBK_EXIT_TIMING": "2"}, {}, {"TBK_EXIT_TIMING": "2"}):
            env = dict(os.environ, TBK_TIMING="1", **dict({"TBK_EXIT_TIMING": "1"}, **extra))
            t0 = time.time()
            r = subprocess.run([exe, "-o", os.path.join(d, "out.bam")] + paths, capture_output=True, text=True, env=env)
            t1 = time.time()
            closed = float(re.search(r"writer closed at ([0-9.]+) ms", r.stderr).group(1))
            rel = float(re.search(r"released the large buffers in ([0-9.]+) ms", r.stderr).group(1))
            m = re.search(r"exit timing: tbk_destroy ([0-9.]+) ms; _exit at ([0-9.]+)", r.stderr)
            x = float(m.group(2))
            print("wall %.3f s | start -> _exit %.3f s (writer closed at %.3f, buffers released in %.3f, tbk_destroy %.1f ms) => before main's clock %.3f s | after _exit %.3f s  %s"
                  % (t1 - t0, x - t0, closed / 1e3, rel / 1e3, float(m.group(1)), (x - t0) - closed / 1e3 - rel / 1e3 - float(m.group(1)) / 1e3, t1 - x, extra), flush=True)
            os.sync()
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
