#!/bin/bash
# the other shapes on the current code: c4 per-rank shape (plain and as a one-rank multi-rank step), c4 as one job, c5 shape
mkdir -p gpurun_out/shapes
C="--no-cpu-baseline --no-host-path --no-e2e"
timeout -k 10 300 python bench.py --profile c4 --scaling weak $C > gpurun_out/shapes/c4.json 2> gpurun_out/shapes/c4.err
timeout -k 10 300 python bench.py --force-dist --profile c4 $C 2> gpurun_out/shapes/c4d.err | tail -n 1 > gpurun_out/shapes/c4d.json
timeout -k 10 400 python bench.py --profile c4 --scaling strong $C > gpurun_out/shapes/c4s.json 2> gpurun_out/shapes/c4s.err
timeout -k 10 300 python bench.py --profile c5 $C > gpurun_out/shapes/c5.json 2> gpurun_out/shapes/c5.err
python - <<P
import json
for n in ("c4","c4d","c4s","c5"):
    try:
        d=json.load(open("gpurun_out/shapes/%s.json"%n))
        print(n, "ms", d["ms_per_step"], "value %.3g"%d["value"], "plain", d.get("plain_ms_per_step"), " ".join("%s=%.2f"%(k.split("/")[-1],v) for k,v in list(d["kernel_ms_per_step"].items())[:9]))
    except Exception as e: print(n, "failed", e)
P
