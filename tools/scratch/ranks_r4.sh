#!/bin/bash
mkdir -p gpurun_out/ranks4
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py -x -q -k "ranks" > gpurun_out/ranks4/pytest.log 2>&1; rc=$?
tail -30 gpurun_out/ranks4/pytest.log
exit $rc
