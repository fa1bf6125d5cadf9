import json, sys
d = json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["gpu_kernel_ms_per_step_total"], d["launches_per_step"])
print({k: v for k, v in d["kernel_ms_per_step"].items() if v > 1.5 or "wg_" in k})
