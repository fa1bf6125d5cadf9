#!/bin/bash
L=gpurun_out/soak_long.log; : > $L
run() { name=$1; shift; echo "== $name $(date +%T)" >> $L; timeout -k 10 $1 python ${@:2} >> $L 2>&1; rc=$?; echo "== $name rc=$rc" | tee -a $L; [ $rc -eq 0 ]; }
run synth 330 tools/scratch/synth_soak.py 2500 &&
run fuzz 330 tools/scratch/fuzz_soak.py 9000 9600 &&
run dist 300 tools/scratch/dist_soak.py 500 &&
run ydlist 200 tools/scratch/yd_list_soak.py 3000
