#!/bin/bash
mkdir -p gpurun_out/h2h4
for cfg in "12 3 1" "12 3 2" "12 4 2" "10 3 2" "12 3 1"; do
set -- $cfg
TBK_H2H_LINK=$3 timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e --steps 4 --warmup 2 --prof-steps 1 --cov-prof-reps 0 --host-subtiles $1 --host-contexts $2 > gpurun_out/h2h4/bench_$1_$2.json 2> gpurun_out/h2h4/bench.err || { tail -20 gpurun_out/h2h4/bench.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/h2h4/bench_$1_$2.json"))
h=d["kernel_path_host_to_host"]
print("$cfg", "h2h", h["ms_per_step"], h["link_gb_per_s"], h["host_wall_ms_summed_over_sub_tiles"])
P
done
