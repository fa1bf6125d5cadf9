#!/bin/bash
# bench.py with 2..3 contexts on c3
for c in 2 3; do
  echo -n "c3 contexts $c: "; python3 bench.py --contexts $c --steps 24 --warmup 4 --prof-steps 1 --no-cpu-baseline --no-host-path | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['hbm_in_use_gb'])"
done
