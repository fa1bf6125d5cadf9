#!/bin/bash
for e in 8 4 16; do
  export TBK_SO2_E=$e
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 --prof-steps 3 > gpurun_out/so2_$e.json 2> gpurun_out/so2_$e.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/so2_$e.json") if l.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("E=$e step %.2f" % d["ms_per_step"], {a:round(b,3) for a,b in k.items() if "chains" in a or "bundles" in a}, d["config"]["groups_out"])
P
done
