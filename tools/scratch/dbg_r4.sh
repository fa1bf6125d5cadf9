#!/bin/bash
mkdir -p gpurun_out/dbg
timeout -k 10 600 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV nostop noprint pass" -ex "run" -ex "bt 30" --args python -m pytest tests/test_gpu_dist.py -x -q -k "test_loopback_device_resident_equals_flat_oracle" > gpurun_out/dbg/gdb.log 2>&1
grep -n "^#\|SIGABRT\|signal" gpurun_out/dbg/gdb.log | head -50 | cut -c1-260
