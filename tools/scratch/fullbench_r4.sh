#!/bin/bash
mkdir -p gpurun_out/full
timeout -k 10 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err || { tail -n 20 gpurun_out/full/bench.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/full/bench.json"))
print("ms_per_step", d["ms_per_step"], "value %.4g" % d["value"], "contexts", d["config"].get("contexts"), "hbm", d["config"].get("hbm_in_use_gb"))
print("roofline", d["roofline"]["avg_launch_us"], d["roofline"]["frac"], "cov", d["roofline_coverage"]["avg_launch_us"])
print("h2h", d["kernel_path_host_to_host"]["ms_per_step"], "e2e", d["end_to_end"]["value"], d["end_to_end_seq"]["value"], d["end_to_end_c3_options"]["value"], "cpu", d["cpu_baseline"]["value"])
P
