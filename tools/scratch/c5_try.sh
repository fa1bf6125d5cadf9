#!/bin/bash
for extra in "--steps 8 --warmup 3" "--steps 8 --warmup 3" "--steps 6 --warmup 2" "--steps 16 --warmup 6"; do
  timeout -k 10 250 python bench.py --profile c5 --no-cpu-baseline --no-host-path --no-e2e $extra > gpurun_out/c5_try.json 2> gpurun_out/c5_try.err
  echo "== $extra"; python tools/scratch/show_bench.py gpurun_out/c5_try.json | grep "ms_per_step"
done
