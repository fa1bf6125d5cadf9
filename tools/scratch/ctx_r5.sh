#!/bin/bash
# the default bench at several numbers of contexts: ms per step, HBM in use
cd "$GRAFT_REPO_ROOT"
for c in "$@"; do
  python3 bench.py --contexts $c --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/ctx_$c.json 2> gpurun_out/ctx_$c.err
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ctx_$c.json') if l.startswith('{')][-1])
print('contexts $c: ms_per_step', d['ms_per_step'], 'value %.2f G' % (d['value']/1e9), {k:v for k,v in d['config'].items() if 'hbm' in k.lower() or 'context' in k.lower()})"
done
