#!/bin/bash
# copies what tools/refresh_profiles.sh (parts a and b) left under gpurun_out/prof/ to the names profiles/README.md lists
set -e
P=gpurun_out/prof; R=${1:-r4}
cp $P/bench_default.json profiles/${R}_bench_c3.json
cp $P/kernel_stats.csv profiles/${R}_bench_c3_kernel_stats.csv
cp $P/pmc_fetch_size.csv profiles/${R}_bench_c3_pmc_fetch_size.csv
cp $P/pmc_write_size.csv profiles/${R}_bench_c3_pmc_write_size.csv
cp $P/kernel_stats_serial.csv profiles/${R}_bench_c3_serial_kernel_stats.csv
cp $P/bench_serial_under_rocprof.json profiles/${R}_bench_c3_serial_under_rocprof.json
cp $P/bench_under_rocprof.json profiles/${R}_bench_c3_under_rocprof.json
cp $P/pmc_sq_wg.txt profiles/${R}_bench_c3_pmc_sq_wg.txt
cp $P/pmc_sq_all.txt profiles/${R}_bench_c3_pmc_sq.txt
cp $P/bench_force_dist.json profiles/${R}_bench_c4shape_force_dist.json
cp $P/bench_force_dist_c3.json profiles/${R}_bench_c3shape_force_dist.json
cp $P/cov_prof_c3.txt profiles/${R}_cov_prof_c3.txt
cp $P/prof_dist_c4shape.txt profiles/${R}_prof_dist_c4shape.txt
cp $P/prof_dist_c3shape.txt profiles/${R}_prof_dist_c3shape.txt
cp $P/traffic.json profiles/traffic_c3_64x5000000.json
cp $P/bench_c4_strong_n1.json profiles/${R}_bench_c4_strong_n1.json
cp $P/bench_c4shape.json profiles/${R}_bench_c4shape.json
cp $P/bench_c5shape.json profiles/${R}_bench_c5shape.json
cp $P/cov_prof_c3_intervals_only.txt profiles/${R}_cov_prof_c3_intervals_only.txt
# round 5's additions (tools/refresh_profiles_c.sh)
if [ -f $P/deflate_bench.json ]; then
  cp $P/deflate_bench.json profiles/${R}_deflate_bench.json
  cp $P/deflate_kernel_stats.csv profiles/${R}_deflate_kernel_stats.csv
  cp $P/e2e_seq_cli.txt profiles/${R}_e2e_seq_cli.txt
  cp $P/e2e_seq_kernel_stats.csv profiles/${R}_e2e_seq_kernel_stats.csv
  cp $P/e2e_seq_memory_copy_stats.csv profiles/${R}_e2e_seq_memory_copy_stats.csv
  cp $P/e2e_legs.json profiles/${R}_e2e_legs.json
  cp $P/bench_force_dist_emulate8.json profiles/${R}_bench_c4shape_force_dist_emulate8.json
fi
