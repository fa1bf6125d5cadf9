#!/bin/bash
# the one-off soaks in a row on the final code, each under its own limit; a line per soak, progress into gpurun_out/soak_all.log
L=gpurun_out/soak_all.log; : > $L
run() { name=$1; shift; echo "== $name $(date +%T)" >> $L; timeout -k 10 $1 python ${@:2} >> $L 2>&1; rc=$?; echo "== $name rc=$rc" | tee -a $L; [ $rc -eq 0 ]; }
run synth 240 tools/scratch/synth_soak.py 400 &&
run fuzz 300 tools/scratch/fuzz_soak.py 5000 5150 &&
run window 240 tools/scratch/window_soak.py 40 &&
run ydlist 240 tools/scratch/yd_list_soak.py 60 &&
run offsets 240 tools/scratch/offsets_soak.py 30 &&
run deflate 300 tools/scratch/deflate_soak.py 300 &&
run dist 240 tools/scratch/dist_soak.py 120 &&
run sort 200 tools/scratch/sort_soak.py 7000 7100 &&
run tiecov 200 tools/scratch/tiecov_soak.py 40 &&
run cli 200 tools/scratch/cli_soak.py 40
