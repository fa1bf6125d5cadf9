#!/bin/bash
# the other BASELINE shapes through bench.py: a regression check against round 4's lines (c4 per-rank 4.92 ms, c5 per-rank 13.1 ms, c4 as one job 44.2 ms, c2 1.0 ms)
cd "$GRAFT_REPO_ROOT"
show() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$1', 'ms_per_step', d['ms_per_step'], 'value %.2f G' % (d['value']/1e9), d['config'].get('contexts'), d['config'].get('hbm_in_use_gb'))"; }
python3 bench.py --profile c4 --scaling weak --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/s_c4shape.json 2> gpurun_out/s_c4shape.err; show gpurun_out/s_c4shape.json
python3 bench.py --profile c5 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/s_c5shape.json 2> gpurun_out/s_c5shape.err; show gpurun_out/s_c5shape.json
python3 bench.py --profile c2 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/s_c2.json 2> gpurun_out/s_c2.err; show gpurun_out/s_c2.json
python3 bench.py --profile c4 --scaling strong --no-cpu-baseline --no-host-path --no-e2e --steps 8 --warmup 2 > gpurun_out/s_c4strong.json 2> gpurun_out/s_c4strong.err; show gpurun_out/s_c4strong.json
python3 bench.py --force-dist --profile c4 --no-cpu-baseline --no-host-path --no-e2e > gpurun_out/s_fdist.json 2> gpurun_out/s_fdist.err; show gpurun_out/s_fdist.json
