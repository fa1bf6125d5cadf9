#!/bin/bash
mkdir -p gpurun_out/hwq
C="--no-cpu-baseline --no-host-path --no-e2e --steps 20 --warmup 6 --prof-steps 1 --cov-prof-reps 0"
for cfg in "8 3" "16 3" "24 3" "8 4" "16 4" "8 3" "16 3"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 300 python bench.py $C --contexts $2 > gpurun_out/hwq/b.json 2> gpurun_out/hwq/b.err || { tail -n 3 gpurun_out/hwq/b.err; exit 1; }
  python - <<P
import json
d=json.load(open("gpurun_out/hwq/b.json"))
print("queues $1 contexts $2  ms_per_step", d["ms_per_step"], "hbm", d["config"]["hbm_in_use_gb"])
P
done
