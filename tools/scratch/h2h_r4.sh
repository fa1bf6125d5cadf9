#!/bin/bash
mkdir -p gpurun_out/h2h
for cfg in "12 3" "18 3" "24 3" "24 4" "36 4"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-e2e --steps 6 --warmup 2 --prof-steps 1 --cov-prof-reps 0 --host-subtiles $1 --host-contexts $2 > gpurun_out/h2h/b.json 2> gpurun_out/h2h/b.err || { tail -n 5 gpurun_out/h2h/b.err; exit 1; }
  python - <<P
import json
d=json.load(open("gpurun_out/h2h/b.json"))["kernel_path_host_to_host"]
print("subtiles $1 contexts $2  ms", d["ms_per_step"], "link GB/s", d["link_gb_per_s"], d["host_wall_ms_summed_over_sub_tiles"])
P
done
