#!/bin/bash
for g in 0 1; do
  export TBK_BENCH_GATE=$g
  timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 --prof-steps 1 > gpurun_out/gate_$g.json 2> gpurun_out/gate_$g.err || tail -5 gpurun_out/gate_$g.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/gate_$g.json") if l.startswith("{")][-1])
print("gate=$g step %.2f ms  value %.3f G/s" % (d["ms_per_step"], d["value"]/1e9))
P
done
export TBK_BENCH_GATE=1
EXTRA="" bash tools/scratch/timeline.sh 2>&1 | head -12
