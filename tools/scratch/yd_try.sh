#!/bin/bash
for wm in 24 24; do
  export TBK_YD_WAVE_MIN=$wm TBK_YD_DEBUG=1
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 > gpurun_out/yd_try.json 2> gpurun_out/yd_try.err
  grep -m1 "^yd:" gpurun_out/yd_try.err | cut -c1-400
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/yd_try.json") if l.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("wave_min=$wm step %.2f  lane %.2f wave %.2f over %.2f" % (d["ms_per_step"], k.get("collapse/yd_lane",0), k.get("collapse/yd_wave",0), k.get("collapse/yd_run_overflow",0)))
P
done
