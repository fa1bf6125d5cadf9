#!/bin/bash
set -u
D=/tmp/e2e_try; rm -rf $D; mkdir -p $D
python - <<P
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tiebrush_amd import synth, synth_dev
t = synth_dev.tile_to_host(synth_dev.make_tile_device(32, 1000000, "c2", device="cuda:0"))
synth.write_bams_fast(t, "$D/in")
P
B=tiebrush_amd/_build
for mode in "TBK_EXIT_TIMING=1" "TBK_EXIT_TIMING=2"; do
for i in 1 2; do
  t0=$(date +%s.%N)
  env $mode TBK_TIMING=1 $B/tiebrush -o $D/out.bam $D/in*.bam > $D/log.txt 2>&1
  t1=$(date +%s.%N)
  grep -E "writer closed|released|exit timing" $D/log.txt
  python3 -c "print('wall %.3f s, process ended at %.3f' % ($t1 - $t0, $t1))"
done
done
