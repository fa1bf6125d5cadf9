#!/bin/bash
# lays down the end_to_end_seq inputs (32 x 1M reads with SEQ / QUAL), runs the command line with its phase times, then once under
# rocprofv3 (kernel + memory-copy trace) so that the device side of the decode / encode shows up call by call
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
D=/tmp/tbk_e2e_prof
mkdir -p $D gpurun_out/e2e_prof
FILES=${1:-32}; READS=${2:-1000000}
python - <<PY
import os, sys, time
sys.path.insert(0, ".")
from tiebrush_amd import synth, synth_dev
t0 = time.time()
tile = synth_dev.tile_to_host(synth_dev.make_tile_device($FILES, $READS, "c2", device="cuda:0"))
paths = synth.write_bams_fast(tile, "$D/in", seq=True)
print("generated", len(paths), "files in %.1f s" % (time.time() - t0), sum(os.path.getsize(p) for p in paths))
PY
for i in 1; do
  S=$(date +%s.%N)
  TBK_TIMING=2 tiebrush_amd/_build/tiebrush -o $D/out.bam $D/in*.bam 2> $D/err.txt
  E=$(date +%s.%N)
  grep -E "hybrid path|writer closed|device writer|written as|released" $D/err.txt
  python3 -c "print('wall %.3f s' % ($E - $S))"
done
cd /tmp && TBK_EXIT_TIMING=3 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/e2e_prof" -o e2e -- "$GRAFT_REPO_ROOT/tiebrush_amd/_build/tiebrush" -o $D/out2.bam $D/in*.bam > "$GRAFT_REPO_ROOT/gpurun_out/e2e_prof/run.log" 2>&1 || true
cd "$GRAFT_REPO_ROOT"
find gpurun_out/e2e_prof -name "*stats*" | head
python3 tools/e2e_prof_summary.py gpurun_out/e2e_prof
