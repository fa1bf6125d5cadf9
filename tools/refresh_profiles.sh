#!/bin/bash
# Regenerates the round's rocprofv3 evidence for the default bench (config 3) on the GPU box; outputs (small) under gpurun_out/prof/.
# Run through gpurun from the repo root, in two calls (each fits gpurun's limit):  gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh a'  then  '... b'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PART=${1:-all}
O=gpurun_out/prof; [ "$PART" != b ] && rm -rf $O; mkdir -p $O
COMMON="--no-cpu-baseline --no-host-path --no-e2e"
if [ "$PART" != b ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --steps 5 --warmup 2 $COMMON > $O/bench_under_rocprof.json 2> $O/ks.err
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/ks
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 bench.py --steps 2 --warmup 1 --prof-steps 1 $COMMON > /dev/null 2> $O/pf.err
python3 tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
rm -rf $O/pf
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 bench.py --steps 2 --warmup 1 --prof-steps 1 $COMMON > /dev/null 2> $O/pw.err
python3 tools/pmc_summary.py $(find $O/pw -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
rm -rf $O/pw
python3 tools/build_traffic.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/traffic.json "c3: 64 files x 5000000 reads, --clip + tiecov"
python3 bench.py > $O/bench_default.json 2> $O/bench.err
tail -c 600 $O/bench_under_rocprof.json; echo; ls -la $O
fi
if [ "$PART" != a ]; then
# the multi-rank step at world = 1 (group partials; its plain-path reference, wire bytes and the record-shuffle fallback ride in the line)
python3 bench.py --force-dist --profile c4 --no-cpu-baseline --no-host-path --no-e2e > $O/bench_force_dist.json 2> $O/bench_force_dist.err
python3 bench.py --force-dist --no-cpu-baseline --no-host-path --no-e2e > $O/bench_force_dist_c3.json 2> $O/bench_force_dist_c3.err
python3 tools/cov_prof.py c3 64 5000000 10 > $O/cov_prof_c3.txt 2> $O/cov_prof.err
python3 tools/prof_dist.py > $O/prof_dist_c4shape.txt 2> $O/prof_dist.err
# the same kernels with the GPU to themselves (one context, the profiling steps of bench.py are serialised calls): the averages
# of this pass are the ones bench.py's roofline objects quote from HIP events
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kss -o kss -- python3 bench.py --contexts 1 --steps 1 --warmup 0 --prof-steps 6 $COMMON > $O/bench_serial_under_rocprof.json 2> $O/kss.err
cp $(find $O/kss -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial.csv
rm -rf $O/kss
# SQ counters (one pass) of the window kernels
PMC_OUT=$O bash tools/pmc_sq.sh prof_sq wg_ > $O/pmc_sq_wg.txt 2> $O/pmc_sq.err
# ... and of every kernel; the multi-rank step unpipelined on config 3's shape
PMC_OUT=$O bash tools/pmc_sq.sh prof_sq_all _k > $O/pmc_sq_all.txt 2> $O/pmc_sq_all.err
python3 tools/prof_dist.py 64 5000000 c3 > $O/prof_dist_c3shape.txt 2> $O/prof_dist_c3.err
# BASELINE config 4 as ONE job on one GPU (the N = 1 end of the strong-scaling workload bench.py runs at N > 1), and its per-rank shape at N = 8
python3 bench.py --profile c4 --scaling strong --no-cpu-baseline --no-host-path --no-e2e --steps 8 --warmup 2 > $O/bench_c4_strong_n1.json 2> $O/bench_c4_strong_n1.err
python3 bench.py --profile c4 --scaling weak --no-cpu-baseline --no-host-path --no-e2e > $O/bench_c4shape.json 2> $O/bench_c4shape.err
python3 bench.py --profile c5 --no-cpu-baseline --no-host-path --no-e2e > $O/bench_c5shape.json 2> $O/bench_c5shape.err
TBK_PROF_NOJ=1 python3 tools/cov_prof.py c3 64 5000000 10 > $O/cov_prof_c3_intervals_only.txt 2>> $O/cov_prof.err
fi
echo refresh done; ls -la $O
