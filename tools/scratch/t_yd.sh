#!/bin/bash
set -e
timeout -k 10 1000 python -m pytest tests/test_gpu_coverage.py tests/test_gpu_collapse.py tests/test_gpu_fuzz.py tests/test_gpu_window.py -x -q -m gpu > gpurun_out/t_yd.log 2>&1 || { tail -30 gpurun_out/t_yd.log; exit 1; }
tail -2 gpurun_out/t_yd.log
timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/t_yd.json 2> gpurun_out/t_yd.err
python tools/scratch/show_bench.py gpurun_out/t_yd.json | grep -v "roofline \|roofline_collapse\|config"
