#!/bin/bash
for mode in "TBK_X=1" "TBK_NO_OFF32=1" "TBK_X=1" "TBK_NO_OFF32=1"; do
  env $mode timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 --prof-steps 3 > gpurun_out/wg_try.json 2> gpurun_out/wg_try.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/wg_try.json") if l.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("$mode step %.2f  wg_hash %.3f wg_hash2 %.3f wg_sort %.3f" % (d["ms_per_step"], k.get("collapse/wg_hash",0), k.get("collapse/wg_hash2",0), k.get("collapse/wg_sort",0)))
P
done
