#!/bin/bash
# timing experiments on wg_hash_k (throw-away variants of libtbk.so under gpurun_exp/: WG_EXP bit 0 = no table work, bit 1 = no loads)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wgexp; rm -rf $O; mkdir -p $O
cp tiebrush_amd/_build/libtbk.so /tmp/libtbk_orig.so
for e in ${@:-0 1 2 3}; do
  if [ $e = 0 ]; then cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so; else cp gpurun_exp/e$e/libtbk.so tiebrush_amd/_build/libtbk.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$e -o ks -- python3 bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-host-path --no-e2e > $O/bench$e.json 2> $O/bench$e.err
  echo "variant $e rc=$?"
  f=$(find $O/ks$e -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E "wg_hash_k|wg_hash2_k" $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-60,200-
  [ -n "$f" ] && python3 - "$f" <<P
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wg_hash_k' in r['Name'] or 'wg_hash2_k' in r['Name']: print('   ', r['Name'][:40], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
P
  rm -rf $O/ks$e
done
cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so
