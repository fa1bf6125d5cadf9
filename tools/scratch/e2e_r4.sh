#!/bin/bash
mkdir -p gpurun_out/e2e4
timeout -k 10 900 python -m pytest tests/test_gpu_bench_ranks.py -x -q > gpurun_out/e2e4/pytest.log 2>&1; rc=$?
tail -15 gpurun_out/e2e4/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tools/e2e_leg.py > gpurun_out/e2e4/e2e.json 2> gpurun_out/e2e4/e2e.err || { tail -5 gpurun_out/e2e4/e2e.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/e2e4/e2e.json"))
for k in (None,"seq","c3_options"):
    e = d if k is None else d[k]
    print(k, e["value"], e["wall_s"], e["input_bytes_per_record"], e["workload"]); print("   ", e["phases"])
P
