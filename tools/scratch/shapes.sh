#!/bin/bash
for p in c2 c4 c5; do
  timeout -k 10 300 python bench.py --profile $p --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 > gpurun_out/shape_$p.json 2> gpurun_out/shape_$p.err || tail -3 gpurun_out/shape_$p.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/shape_$p.json") if l.startswith("{")][-1])
print("$p step %.3f ms  %.2f G rec/s  groups %d" % (d["ms_per_step"], d["value"]/1e9, d["config"]["groups_out"]), d["config"]["workload"][:60])
P
done
timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/shape_dist.json 2> gpurun_out/shape_dist.err || tail -3 gpurun_out/shape_dist.err
python tools/scratch/show_bench.py gpurun_out/shape_dist.json | sed -n 2,12p | cut -c1-300
