#!/bin/bash
mkdir -p gpurun_out/ctx3
for c in 2 3; do
timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e --no-host-path --steps 21 --warmup 6 --prof-steps 1 --cov-prof-reps 0 --contexts $c > gpurun_out/ctx3/bench_$c.json 2> gpurun_out/ctx3/bench.err || { tail -20 gpurun_out/ctx3/bench.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/ctx3/bench_$c.json"))
print("contexts $c", "step", d["ms_per_step"], "hbm", d["config"]["hbm_in_use_gb"])
P
done
TBK_BENCH_GATE=0 timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e --no-host-path --steps 21 --warmup 6 --prof-steps 1 --cov-prof-reps 0 --contexts 3 > gpurun_out/ctx3/bench_3ng.json 2> gpurun_out/ctx3/bench.err
python - <<P
import json
d=json.load(open("gpurun_out/ctx3/bench_3ng.json"))
print("contexts 3 no gate", "step", d["ms_per_step"], "hbm", d["config"]["hbm_in_use_gb"])
P
