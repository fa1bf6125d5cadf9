#!/bin/bash
# wg_hash_k under variants of libtbk.so (gpurun_exp/<name>/libtbk.so, tools/scratch/wgopt_build.sh): serialised kernel time from rocprofv3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wgopt; rm -rf $O; mkdir -p $O
cp tiebrush_amd/_build/libtbk.so /tmp/libtbk_orig.so
for e in "$@"; do
  if [ $e = cur ]; then cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so; else cp gpurun_exp/$e/libtbk.so tiebrush_amd/_build/libtbk.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$e -o ks -- python3 bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-host-path --no-e2e > $O/bench$e.json 2> $O/bench$e.err
  echo "variant $e rc=$?"
  f=$(find $O/ks$e -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<P
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wg_hash_k' in r['Name'] or 'wg_hash2_k' in r['Name'] or 'wg_sort_k' in r['Name']: print('   ', r['Name'][:60].replace('(anonymous namespace)::',''), r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
P
  python3 -c "
import json,sys
l=[x for x in open('$O/bench$e.json') if x.startswith('{')]
d=json.loads(l[-1]); print('    ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['achieved'])"
  rm -rf $O/ks$e
done
cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so
