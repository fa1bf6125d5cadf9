#!/bin/bash
for p in c4 c5 c2; do
  timeout -k 10 250 python bench.py --profile $p --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/shape_$p.json 2> gpurun_out/shape_$p.err
  echo "== $p"; python tools/scratch/show_bench.py gpurun_out/shape_$p.json | grep "value\|ms_per_step\|kernels" | cut -c1-260
done
