#!/bin/bash
mkdir -p gpurun_out/cov4
for xp in 0 1 2 4 8 16 32 63; do
  echo "== xp $xp"
  TBK_CL_XP=$xp TBK_PROF_NOJ=1 timeout -k 10 300 python tools/cov_prof.py 2>&1 | grep -E "coverage call|cov_place|cov_bundles"
done
