#!/bin/bash
# other single-GPU shapes through bench.py (config 2, config 4's per-rank shape, config 5's per-rank shape): ms per step and top kernels
mkdir -p gpurun_out/qp
for p in "$@"; do
  timeout -k 10 300 python bench.py --profile $p --no-cpu-baseline --no-host-path --steps 20 > gpurun_out/qp/$p.json 2> gpurun_out/qp/$p.err || { tail -3 gpurun_out/qp/$p.err; continue; }
  python - <<P
import json
d=json.load(open("gpurun_out/qp/$p.json"))
print("$p", "ms_per_step", d["ms_per_step"], "value %.3g" % d["value"], "kernel_sum", d["gpu_kernel_ms_per_step_total"], "launches", d["launches_per_step"])
print("   "+" ".join("%s=%.3f" % (k.split("/")[1] if k.startswith("collapse") else k, v) for k, v in list(d["kernel_ms_per_step"].items())[:14]))
P
done
