#!/usr/bin/env python3
"""the hybrid decode's split: wall time of the `tiebrush` command line on the end_to_end_seq inputs for several device shares
(TBK_HYBRID_SHARE, per cent of the compressed bytes) — what the constants of the split rule in tiebrush_main.cpp are fitted to"""
import os, re, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from tiebrush_amd import synth, synth_dev
    files, reads = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    shares = [s for s in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["default", "45", "55", "60", "65", "70"])]
    d = tempfile.mkdtemp(prefix="tbk_share_", dir="/tmp")
    try:
        tile = synth_dev.tile_to_host(synth_dev.make_tile_device(files, reads, "c2", device="cuda:0"))
        torch.cuda.empty_cache()
        paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=True)
        del tile
        os.sync()
        exe = os.path.join(ROOT, "tiebrush_amd", "_build", "tiebrush")
        for sh in shares:
            env = dict(os.environ, TBK_TIMING="1")
            if sh != "default":
                env["TBK_HYBRID_SHARE"] = sh
            ts, line = [], ""
            for _ in range(3):
                t0 = time.time()
                r = subprocess.run([exe, "-o", os.path.join(d, "out.bam")] + paths, capture_output=True, text=True, env=env, check=True)
                ts.append(time.time() - t0)
                line = next((l for l in r.stderr.split("\n") if l.startswith("hybrid path ms")), "")
                os.sync()
            m = re.search(r"device (\d+) of (\d+) files \(context ready at ([0-9.]+) .*?decode incl. context ([0-9.]+), the call ([0-9.]+)\) beside host \(.*?\) = ([0-9.]+)", line)
            print("share %-7s wall median %.3f min %.3f | %s" % (sh, sorted(ts)[1], min(ts), "device %s of %s files, ctx %s, device side %s (call %s), phase %s" % m.groups() if m else line[:120]), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
