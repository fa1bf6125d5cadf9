#!/bin/bash
# round 4: coverage chain check — coverage tests, then the per-kernel table of the call on config 3
mkdir -p gpurun_out/cov4
timeout -k 10 600 python -m pytest tests/test_gpu_coverage.py tests/test_gpu_collapse.py tests/test_gpu_cli.py -x -q > gpurun_out/cov4/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/cov4/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/cov_prof.py > gpurun_out/cov4/cov_prof.txt 2>&1 || { tail -5 gpurun_out/cov4/cov_prof.txt; exit 1; }
cat gpurun_out/cov4/cov_prof.txt
TBK_COV_LEGACY=1 timeout -k 10 300 python tools/cov_prof.py > gpurun_out/cov4/cov_prof_legacy.txt 2>&1 && head -3 gpurun_out/cov4/cov_prof_legacy.txt
TBK_PROF_NOJ=1 timeout -k 10 300 python tools/cov_prof.py > gpurun_out/cov4/cov_prof_noj.txt 2>&1 && head -12 gpurun_out/cov4/cov_prof_noj.txt
