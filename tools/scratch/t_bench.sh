#!/bin/bash
# bench only (no tests): quick A/B of a kernel change
timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/t_win.json 2> gpurun_out/t_win.err
python tools/scratch/show_bench.py gpurun_out/t_win.json | grep -v roofline
