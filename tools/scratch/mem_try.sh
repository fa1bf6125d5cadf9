#!/bin/bash
export TBK_WS_DEBUG=1
for p in c3 c5 c4; do
  timeout -k 10 300 python bench.py --profile $p --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 > gpurun_out/mem_$p.json 2> gpurun_out/mem_$p.err || tail -3 gpurun_out/mem_$p.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/mem_$p.json") if l.startswith("{")][-1])
print("$p step %.3f ms  hbm %.1f GB" % (d["ms_per_step"], d["config"]["hbm_in_use_gb"]))
P
  grep "overflow" gpurun_out/mem_$p.err | grep -v "+ 0.00 GB" | sort | uniq -c | head -4
done
unset TBK_WS_DEBUG
timeout -k 10 600 python bench.py --force-dist --profile c3 --steps 6 --warmup 2 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/dist_c3.json 2> gpurun_out/dist_c3.err || tail -3 gpurun_out/dist_c3.err
python tools/scratch/show_bench.py gpurun_out/dist_c3.json | sed -n 2,10p | cut -c1-200
