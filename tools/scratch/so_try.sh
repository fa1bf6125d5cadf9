#!/bin/bash
timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 --prof-steps 3 > gpurun_out/so_try.json 2> gpurun_out/so_try.err
python - <<P
import json
d=json.loads([l for l in open("gpurun_out/so_try.json") if l.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("step %.2f" % d["ms_per_step"], {a:round(b,3) for a,b in k.items() if "chains" in a or "bundles" in a or "scan" in a})
P
