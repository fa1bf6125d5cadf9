#!/bin/bash
for c in 1 2 3; do
  timeout -k 10 200 python bench.py --contexts $c --steps 9 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 --prof-steps 1 > gpurun_out/ctx_$c.json 2> gpurun_out/ctx_$c.err || tail -3 gpurun_out/ctx_$c.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/ctx_$c.json") if l.startswith("{")][-1])
print("contexts=$c step %.2f ms  hbm %.0f GB" % (d["ms_per_step"], d["config"]["hbm_in_use_gb"]))
P
done
