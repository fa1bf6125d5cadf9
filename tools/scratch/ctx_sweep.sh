#!/bin/bash
# bench.py with 1..4 contexts: c3 (memory allows 2) and c2
for c in 1 2; do
  echo -n "c3 contexts $c: "; python3 bench.py --contexts $c --steps 24 --warmup 4 --prof-steps 0 --no-cpu-baseline --no-host-path | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
for c in 1 2 3 4; do
  echo -n "c2 contexts $c: "; python3 bench.py --profile c2 --contexts $c --steps 400 --warmup 20 --prof-steps 0 --no-cpu-baseline --no-host-path | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
for c in 2 3 4; do
  echo -n "c3-shaped 64x2M contexts $c: "; python3 bench.py --reads-per-file 2000000 --contexts $c --steps 24 --warmup 4 --prof-steps 0 --no-cpu-baseline --no-host-path | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
