#!/bin/bash
mkdir -p gpurun_out/inf
timeout -k 10 400 python -m pytest tests/test_gpu_bgzf.py -x -q -m gpu > gpurun_out/inf/pytest.log 2>&1
rc=$?
tail -n 3 gpurun_out/inf/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 bash tools/scratch/dd2_r4.sh > gpurun_out/inf/dd_wave.log 2>&1 && grep -E "bam_decode wall|bgz_|bam_index" gpurun_out/inf/dd_wave.log
