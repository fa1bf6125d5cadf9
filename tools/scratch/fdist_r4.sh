#!/bin/bash
mkdir -p gpurun_out/fdist
C="--no-cpu-baseline --no-host-path --no-e2e"
for p in c4 c4 c3; do
timeout -k 10 300 python bench.py --force-dist --profile $p $C 2> gpurun_out/fdist/b.err | tail -n 1 > gpurun_out/fdist/b.json || { tail -n 5 gpurun_out/fdist/b.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/fdist/b.json"))
print("$p dist ms", d["ms_per_step"], "plain", d.get("plain_ms_per_step"), "ratio %.3f" % (d["ms_per_step"]/d["plain_ms_per_step"]), "hbm", d["config"].get("hbm_in_use_gb"))
P
done
