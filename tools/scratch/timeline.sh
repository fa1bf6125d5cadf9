#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 12 --warmup 3 --prof-steps 1 --cov-prof-reps 0 --no-cpu-baseline --no-host-path --no-e2e ${EXTRA:-} > gpurun_out/tl_bench.json 2> gpurun_out/tl.err
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 tools/scratch/timeline.py $f
python3 tools/scratch/show_bench.py gpurun_out/tl_bench.json | sed -n 2,3p
rm -rf gpurun_out/tl
