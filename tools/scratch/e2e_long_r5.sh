#!/bin/bash
# the long end-to-end leg: FILES x READS reads with SEQ / QUAL through the command line at several device shares of the hybrid decode
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
D=/tmp/tbk_e2e_long
mkdir -p $D
FILES=${1:-64}; READS=${2:-2000000}; shift; shift
python - <<PY
import os, sys, time
sys.path.insert(0, ".")
from tiebrush_amd import synth, synth_dev
t0 = time.time()
tile = synth_dev.tile_to_host(synth_dev.make_tile_device($FILES, $READS, "c2", device="cuda:0"))
print("tile on the host in %.1f s" % (time.time() - t0)); sys.stdout.flush()
paths = synth.write_bams_fast(tile, "$D/in", seq=True)
print("generated", len(paths), "files in %.1f s" % (time.time() - t0), sum(os.path.getsize(p) for p in paths))
PY
for share in "$@"; do
  for i in 1 2 3; do
    S=$(date +%s.%N)
    TBK_HYBRID_SHARE=$share TBK_TIMING=1 tiebrush_amd/_build/tiebrush -o $D/out.bam $D/in*.bam 2> $D/err.txt
    E=$(date +%s.%N)
    grep -E "hybrid|host path|writer closed|device writer|written as|released|tiles|device decode kernels" $D/err.txt
    python3 -c "print('share $share: wall %.3f s' % ($E - $S))"
  done
done
ls -la $D/out.bam
rm -rf $D
