#!/bin/bash
mkdir -p gpurun_out/shapes4
run() { # name args...
  n=$1; shift
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e --no-host-path --prof-steps 1 --cov-prof-reps 0 "$@" > gpurun_out/shapes4/$n.json 2> gpurun_out/shapes4/$n.err || { echo "$n FAILED"; tail -5 gpurun_out/shapes4/$n.err; return; }
  python - <<P
import json
d=json.load(open("gpurun_out/shapes4/$n.json"))
print("$n", "ms", d["ms_per_step"], "value %.3g" % d["value"], "hbm", d["config"]["hbm_in_use_gb"], d["config"]["workload"][:60], "scaling", d["scaling"])
P
}
run c3 --steps 10 --warmup 3
run c4strong1 --profile c4 --scaling strong --steps 6 --warmup 2
run c4weak --profile c4 --steps 20 --warmup 4
run c5 --profile c5 --steps 10 --warmup 3
run c2 --profile c2 --steps 40 --warmup 5
