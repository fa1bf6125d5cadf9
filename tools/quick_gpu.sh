#!/bin/bash
# quick GPU check used during kernel work: window-path tests, then a short default bench (config 3) with the per-kernel table
# usage (through gpurun): bash tools/quick_gpu.sh <tag> [pytest targets...]
tag=${1:-q}; shift
mkdir -p gpurun_out/$tag
targets=${@:-tests/test_gpu_window.py}
timeout -k 10 900 python -m pytest $targets -x -q > gpurun_out/$tag/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/$tag/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline --no-host-path --steps 10 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -5 gpurun_out/$tag/bench.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "kernel_sum", d["gpu_kernel_ms_per_step_total"], "launches", d["launches_per_step"])
print(" ".join("%s=%.2f" % (k.split("/")[1] if k.startswith("collapse") else k, v) for k, v in list(d["kernel_ms_per_step"].items())[:22]))
P
