#!/bin/bash
# where the collapse call of the hybrid command line goes (32 x 1 M reads with SEQ / QUAL)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
D=/tmp/tbk_e2ecol; mkdir -p $D
python - <<PY
import os, sys, time
sys.path.insert(0, ".")
from tiebrush_amd import synth, synth_dev
tile = synth_dev.tile_to_host(synth_dev.make_tile_device(32, 1000000, "c2", device="cuda:0"))
paths = synth.write_bams_fast(tile, "$D/in", seq=True)
os.sync()
PY
for i in 1 2 3; do
  S=$(date +%s.%N)
  TBK_TIMING=1 ${TBK_PRE:-} tiebrush_amd/_build/tiebrush -o $D/out.bam $D/in*.bam 2> $D/err.txt
  E=$(date +%s.%N)
  grep -E "hybrid path|collapse call|device decode kernels" $D/err.txt | cut -c1-420
  python3 -c "print('wall %.3f s' % ($E - $S))"
done
rm -rf $D
