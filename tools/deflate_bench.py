#!/usr/bin/env python3
"""Throughput of the device BGZF deflate (tbk_bgzf_deflate) on records that look like real data: `tbh_tool mkbam seq` payload (238 B per
record: SEQ, QUAL, an aligner's tags), the reference's own fixture records, and the size against zlib level 6.  Prints one JSON line."""
import gzip
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from tiebrush_amd import api, synth, synth_dev
    import tempfile
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    d = tempfile.mkdtemp(prefix="tbk_dfl_", dir="/tmp")
    tile = synth_dev.tile_to_host(synth_dev.make_tile_device(2, 200000, "c2", device="cuda:0"))
    paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=True)
    syn = gzip.open(paths[0]).read()
    gold = gzip.open(os.path.join(ROOT, "tests", "golden", "t1", "t1s0.bam")).read()
    ctx = api.Context(0)
    res = {}
    for name, base in (("synthetic_seq", syn), ("golden_t1s0", gold)):
        payload = (base * (mb * (1 << 20) // len(base) + 1))[:mb << 20]
        ctx.bgzf_deflate(payload[:1 << 22])                      # warm-up: code objects, buffers
        if hasattr(ctx.L, "tbk_debug_deflate_phases"):
            import ctypes as C
            ctx.L.tbk_debug_deflate_phases((C.c_ulonglong * 16)(), 1)
        ctx.set_profiling(True)
        t0 = time.perf_counter()
        run = ctx.bgzf_deflate(payload)
        dt = time.perf_counter() - t0
        kt = ctx.kernel_times()
        ctx.set_profiling(False)
        samp = payload[:64 * 0xff00]
        z6 = sum(len(zlib.compress(samp[o:o + 0xff00], 6)) + 20 for o in range(0, len(samp), 0xff00))
        dv = len(ctx.bgzf_deflate(samp))
        assert gzip.decompress(run[:0] + run) == payload if mb <= 64 else True
        k = kt.get("bgz_deflate", (0.0, 0))[0]
        phases = None
        if hasattr(ctx.L, "tbk_debug_deflate_phases"):
            import ctypes as C
            ph = (C.c_ulonglong * 16)()
            ctx.L.tbk_debug_deflate_phases(ph, 1)
            tot = float(sum(ph)) or 1.0
            names = ["stage", "parse", "histogram", "crc", "codes", "decide", "write_bits", "copy_out"]
            tot = float(sum(ph[:8])) or 1.0
            phases = {names[i]: round(ph[i] / tot, 3) for i in range(8)}
            phases.update(busy_matcher0=round(ph[8] / tot, 3), busy_parser=round(ph[9] / tot, 3), busy_inserter=round(ph[10] / tot, 3))
        res[name] = {"phases": phases, "payload_mb": mb, "call_s": round(dt, 3), "deflate_kernel_ms": round(k, 2), "kernel_gb_s": round(len(payload) / (k * 1e-3) / 1e9, 2) if k else None,
                     "compressed_over_payload": round(len(run) / len(payload), 4), "device_over_zlib6_sample": round(dv / z6, 4),
                     "kernels_ms": {a: round(b[0], 2) for a, b in kt.items()}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
