#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ls
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ls -o ls -- python3 tools/launch_seq.py run c2 2 1000000 > gpurun_out/ls_run.log 2>&1
f=$(find gpurun_out/ls -name "*kernel_trace.csv" | head -1)
python3 tools/launch_seq.py show $f > gpurun_out/ls_c2.txt
tail -3 gpurun_out/ls_c2.txt
python3 bench.py --profile c2 --contexts 1 --steps 200 --warmup 20 --no-cpu-baseline --no-host-path --no-e2e --cov-prof-reps 0 > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err || tail -5 gpurun_out/c2_bench.err
python3 tools/scratch/show_bench.py gpurun_out/c2_bench.json | head -8
