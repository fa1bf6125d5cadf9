#!/bin/bash
# usage (gpurun): bash tools/scratch/e2e_try.sh  -> phases of the tiebrush command line on 32 x 1M reads
set -u
D=/tmp/e2e_try; rm -rf $D; mkdir -p $D
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
python - <<P
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tiebrush_amd import synth, synth_dev
t = synth_dev.tile_to_host(synth_dev.make_tile_device(32, 1000000, "c2", device="cuda:0"))
synth.write_bams_fast(t, "$D/in")
P
B=tiebrush_amd/_build
for mode in "TBK_X=0" "TBK_X=0" "TBK_THREADS=32" "TBK_BAM_LEVEL=1"; do
  echo "== $mode"
  for i in 1 2; do
    t0=$(date +%s.%N)
    env $mode TBK_TIMING=1 $B/tiebrush -o $D/out.bam $D/in*.bam > $D/log.txt 2>&1
    t1=$(date +%s.%N)
    grep -E "timing|device|host path|writer closed|Error" $D/log.txt
    python3 -c "print(\"wall %.3f s\" % ($t1 - $t0))"
  done
done
