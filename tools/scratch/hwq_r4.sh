#!/bin/bash
# step-time spread of the default bench, and whether more hardware queues (GPU_MAX_HW_QUEUES) change it
mkdir -p gpurun_out/hwq
C="--no-cpu-baseline --no-host-path --no-e2e --steps 20"
for q in default 8 default 8 2 default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 200 python bench.py $C > gpurun_out/hwq/b.json 2> gpurun_out/hwq/b.err || { tail -n 3 gpurun_out/hwq/b.err; exit 1; }
  python - <<P
import json
d=json.load(open("gpurun_out/hwq/b.json")); k=d["kernel_ms_per_step"]
print("queues $q  ms_per_step", d["ms_per_step"], "kernel_sum", d["gpu_kernel_ms_per_step_total"], "yd_lane", k.get("collapse/yd_lane"), "yd_wave", k.get("collapse/yd_wave"), "wg_hash", k.get("collapse/wg_hash"))
P
done
