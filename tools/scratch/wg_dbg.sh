#!/bin/bash
export TBK_WG_DEBUG=1
timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 --prof-steps 1 > gpurun_out/wg_dbg.json 2> gpurun_out/wg_dbg.err
grep -m 24 "wg_hash phase\|wg_window\|wg windows" gpurun_out/wg_dbg.err
