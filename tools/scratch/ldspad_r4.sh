#!/bin/bash
mkdir -p gpurun_out/pad
C="--no-cpu-baseline --no-host-path --no-e2e --steps 20"
for pad in 0 12800 0 12800; do
  export TBK_WG_LDS_PAD=$pad
  timeout -k 10 200 python bench.py $C > gpurun_out/pad/b.json 2> gpurun_out/pad/b.err || { tail -n 3 gpurun_out/pad/b.err; exit 1; }
  python - <<P
import json
d=json.load(open("gpurun_out/pad/b.json")); k=d["kernel_ms_per_step"]
print("pad $pad  ms_per_step", d["ms_per_step"], "kernel_sum", d["gpu_kernel_ms_per_step_total"], "yd_lane", k.get("collapse/yd_lane"), "yd_wave", k.get("collapse/yd_wave"), "wg_hash", k.get("collapse/wg_hash"))
P
done
