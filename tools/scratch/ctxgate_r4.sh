#!/bin/bash
# contexts (2, 3) x the one-collapse-at-a-time gate (on, off), twenty steps each, twice
mkdir -p gpurun_out/ctxg
for rep in 1 2; do
for c in 2 3; do for g in 1 0; do
TBK_BENCH_GATE=$g timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --no-host-path --steps 20 --warmup 6 --prof-steps 1 --cov-prof-reps 0 --contexts $c > gpurun_out/ctxg/b.json 2> gpurun_out/ctxg/b.err || { tail -5 gpurun_out/ctxg/b.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/ctxg/b.json"))
print("contexts $c gate $g", "step", d["ms_per_step"], "hbm", d["config"]["hbm_in_use_gb"])
P
done; done; done
