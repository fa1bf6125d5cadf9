#!/bin/bash
# stream priorities of the side contexts (gpurun_exp/prio<m>: 0 none, 1 the junction branch's, 2 the YD stage's; the tree's build: both) at 3 and 4 contexts
cd "$GRAFT_REPO_ROOT"
cp tiebrush_amd/_build/libtbk.so /tmp/libtbk_orig.so
for e in "$@"; do
  if [ $e = cur ]; then cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so; else cp gpurun_exp/$e/libtbk.so tiebrush_amd/_build/libtbk.so; fi
  for c in 3 4 3; do
    python3 bench.py --contexts $c --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/prio_$e_$c.json 2> gpurun_out/prio_$e_$c.err
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/prio_$e_$c.json') if l.startswith('{')][-1])
print('$e contexts $c: ms_per_step', d['ms_per_step'], 'cov call', d['roofline_coverage'].get('coverage_call_ms_median'))"
  done
done
cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so
