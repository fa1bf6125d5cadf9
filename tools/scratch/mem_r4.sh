#!/bin/bash
mkdir -p gpurun_out/mem4
TBK_WS_DEBUG=1 timeout -k 10 600 python bench.py --no-cpu-baseline --no-host-path --no-e2e --steps 6 --warmup 2 > gpurun_out/mem4/bench.json 2> gpurun_out/mem4/bench.err
grep "tbk arena" gpurun_out/mem4/bench.err | sort | uniq -c | sort -rn | head -20
python - <<P
import json
d=json.load(open("gpurun_out/mem4/bench.json"))
print("ms", d["ms_per_step"], "hbm", d["config"]["hbm_in_use_gb"])
P
timeout -k 10 600 python tools/e2e_leg.py --no-extra > /dev/null 2>&1
python - <<P
import sys,json,subprocess
sys.path.insert(0,".")
sys.argv=["x"]
import tools.e2e_leg as e
r=e.leg(32,1000000,"c2",[],True,2,True,"%d x %d seq")
print(json.dumps(r,indent=1))
P
