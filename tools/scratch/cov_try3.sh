#!/bin/bash
set -e
timeout -k 10 900 python -m pytest tests/test_gpu_coverage.py tests/test_gpu_fuzz.py tests/test_gpu_collapse.py tests/test_gpu_window.py -x -q -m gpu > gpurun_out/cov_try_tests.log 2>&1 || { tail -30 gpurun_out/cov_try_tests.log; exit 1; }
tail -2 gpurun_out/cov_try_tests.log
timeout -k 10 200 python tools/cov_prof.py c3 64 5000000 8 2>/dev/null | sed -n '2,16p'
timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/cov_try3.json 2> gpurun_out/cov_try3.err
python tools/scratch/show_bench.py gpurun_out/cov_try3.json
