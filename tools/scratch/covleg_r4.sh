#!/bin/bash
mkdir -p gpurun_out/covleg
TBK_PROF_NOJ=1 TBK_COV_LEGACY=1 timeout -k 10 300 python3 tools/cov_prof.py c3 64 5000000 10 > gpurun_out/covleg/legacy.txt 2> gpurun_out/covleg/err.txt
cat gpurun_out/covleg/legacy.txt
