#!/bin/bash
# the streamed (bounded-memory) path of the command line on FILES x READS reads with SEQ / QUAL at several tile sizes, beside the whole-input path
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
D=/tmp/tbk_stream
mkdir -p $D
FILES=${1:-32}; READS=${2:-1000000}; shift; shift
python - <<PY
import os, sys, time
sys.path.insert(0, ".")
from tiebrush_amd import synth, synth_dev
t0 = time.time()
tile = synth_dev.tile_to_host(synth_dev.make_tile_device($FILES, $READS, "c2", device="cuda:0"))
paths = synth.write_bams_fast(tile, "$D/in", seq=True)
print("generated", len(paths), "files in %.1f s" % (time.time() - t0), sum(os.path.getsize(p) for p in paths))
os.sync()
PY
run() {  # name, env...
  local name=$1; shift
  for i in 1 2; do
    S=$(date +%s.%N)
    env "$@" TBK_TIMING=1 tiebrush_amd/_build/tiebrush $WARGS -o $D/out.bam $D/in*.bam 2> $D/err.txt || { cat $D/err.txt; exit 1; }
    mv $D/out.bam $D/$name.bam
    E=$(date +%s.%N)
    grep -E "streamed|device writer|tiles:|hybrid" $D/err.txt
    python3 -c "print('$name: wall %.3f s' % ($E - $S))"
    sync
  done
}
run whole TBK_NOP=1
for t in "$@"; do
  run tile$t TBK_TILE_RECORDS=$t
  WARGS="--writer host" run tile${t}host TBK_TILE_RECORDS=$t
done
python - <<PY
import sys, glob, zlib, struct, hashlib
sys.path.insert(0, ".")
from tiebrush_amd import bamio
def stream(p):
    return hashlib.sha256(bamio.bgzf_decompress(open(p, "rb").read())).hexdigest()
ref = stream("$D/whole.bam")
for p in sorted(glob.glob("$D/tile*.bam")):
    print(p, "record stream equals the whole-input run:", stream(p) == ref)
PY
ls -la $D/*.bam | grep -v "/in"
rm -rf $D
