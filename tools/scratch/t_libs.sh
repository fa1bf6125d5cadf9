#!/bin/bash
# A/B of library variants on one box: swaps _build/libtbk.so for each variant (the original is restored at the end)
cd tiebrush_amd/_build && cp libtbk.so libtbk_orig.so && cd ../..
for v in ${VARIANTS:-orig p1 p2 p4 orig}; do
  cp tiebrush_amd/_build/libtbk_$v.so tiebrush_amd/_build/libtbk.so
  timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/t_lib_$v.json 2> gpurun_out/t_lib.err
  echo "== $v"; python tools/scratch/show_bench.py gpurun_out/t_lib_$v.json | grep "ms_per_step\|kernels" | cut -c1-120
done
cp tiebrush_amd/_build/libtbk_orig.so tiebrush_amd/_build/libtbk.so
