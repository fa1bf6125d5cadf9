#!/bin/bash
for m in 0 1; do
  if [ $m = 1 ]; then export TBK_WG_PACKED=1; else unset TBK_WG_PACKED; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 --prof-steps 3 > gpurun_out/pk_$m.json 2> gpurun_out/pk_$m.err || tail -3 gpurun_out/pk_$m.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/pk_$m.json") if l.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("packed=$m step %.2f" % d["ms_per_step"], {a:round(b,3) for a,b in k.items() if "wg_hash" in a or "pack" in a}, d["config"]["groups_out"])
P
done
