#!/bin/bash
bash tools/scratch/h2h_r4.sh
mkdir -p gpurun_out/c4full
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py -x -q -k "config4_full" > gpurun_out/c4full/pytest.log 2>&1; tail -15 gpurun_out/c4full/pytest.log
