#!/bin/bash
# A/B of an env hook on the default bench, same box: usage t_ab.sh VAR
for v in "" "$1=1" "" "$1=1"; do
  env $v timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/t_ab.json 2> gpurun_out/t_ab.err
  echo "== [$v]"; python tools/scratch/show_bench.py gpurun_out/t_ab.json | grep "ms_per_step\|kernels"
done
