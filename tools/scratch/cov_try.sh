#!/bin/bash
# coverage call: tests of the junction path, then the default bench's coverage numbers (with and without the block-level sums)
set -e
timeout -k 10 500 python -m pytest tests/test_gpu_coverage.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/cov_try_tests.log 2>&1 || { tail -30 gpurun_out/cov_try_tests.log; exit 1; }
tail -2 gpurun_out/cov_try_tests.log
for agg in 0 1; do
  if [ $agg = 0 ]; then export TBK_NO_JUNC_AGG=1; else unset TBK_NO_JUNC_AGG; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/cov_try_$agg.json 2> gpurun_out/cov_try_$agg.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/cov_try_$agg.json") if l.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("agg=$agg step %.2f" % d["ms_per_step"], {a:round(b,3) for a,b in k.items() if a.startswith("coverage/")}, {a:b for a,b in d.items() if "coverage_call" in a})
P
done
