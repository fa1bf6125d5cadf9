#!/bin/bash
mkdir -p gpurun_out/t1
timeout -k 10 600 python -m pytest "$@" -x -q > gpurun_out/t1/pytest.log 2>&1; rc=$?
tail -n 15 gpurun_out/t1/pytest.log
exit $rc
