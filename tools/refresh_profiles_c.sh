#!/bin/bash
# Round 5's additions to the evidence under profiles/ (the output side and the command line on the device): run through gpurun after parts a and b
# of tools/refresh_profiles.sh; leaves its files under gpurun_out/prof/ as they do.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
# the device deflate alone: throughput and size against zlib level 6 (one JSON line), then its kernels
python3 tools/deflate_bench.py 512 > $O/deflate_bench.json 2> $O/deflate_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dk -o dk -- python3 tools/deflate_bench.py 256 > /dev/null 2> $O/dk.err
cp $(find $O/dk -name "*kernel_stats.csv" | head -1) $O/deflate_kernel_stats.csv
rm -rf $O/dk
# the command line on 32 x 1 M reads with SEQ / QUAL: phase lines, then kernels and copies of one run under rocprofv3
rm -rf gpurun_out/e2e_prof
bash tools/e2e_prof.sh 32 1000000 > $O/e2e_seq_cli.txt 2>&1
cp gpurun_out/e2e_prof/e2e_kernel_stats.csv $O/e2e_seq_kernel_stats.csv
cp gpurun_out/e2e_prof/e2e_memory_copy_stats.csv $O/e2e_seq_memory_copy_stats.csv
rm -rf gpurun_out/e2e_prof /tmp/tbk_e2e_prof
# the end-to-end legs as bench.py runs them (child processes), alone
python3 tools/e2e_leg.py > $O/e2e_legs.json 2> $O/e2e_legs.err || true
# the protocol on config 4's per-rank shape: eight emulated ranks, lists against rounds
python3 bench.py --force-dist --profile c4 --emulate-world 8 --no-cpu-baseline --no-host-path --no-e2e > $O/bench_force_dist_emulate8.json 2> $O/bench_force_dist_emulate8.err
echo refresh c done; ls -la $O
