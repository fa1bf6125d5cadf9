#!/bin/bash
# round-6 GPU check: the whole GPU suite, a short default bench with its per-kernel table, then the two traffic passes
# usage (through gpurun): bash tools/r6_check.sh <tag> [notests] [nopmc]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r6}; O=gpurun_out/$tag; mkdir -p $O
if [ "${2:-}" != notests ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
fi
COMMON="--no-cpu-baseline --no-host-path --no-e2e"
timeout -k 10 400 python bench.py $COMMON --steps 10 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<P
import json
d=json.load(open("$O/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "kernel_sum", d["gpu_kernel_ms_per_step_total"], "launches", d["launches_per_step"])
print(" ".join("%s=%.2f" % (k.split("/")[1] if k.startswith("collapse") else k, v) for k, v in list(d["kernel_ms_per_step"].items())[:30]))
P
if [ "${3:-}" != nopmc ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 bench.py --steps 2 --warmup 1 --prof-steps 1 $COMMON > /dev/null 2> $O/pf.err
python3 tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
rm -rf $O/pf
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 bench.py --steps 2 --warmup 1 --prof-steps 1 $COMMON > /dev/null 2> $O/pw.err
python3 tools/pmc_summary.py $(find $O/pw -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
rm -rf $O/pw
python3 tools/build_traffic.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/traffic.json "c3: 64 files x 5000000 reads, --clip + tiecov"
python3 - <<P
import json
b=json.load(open("$O/traffic.json"))["bytes_per_launch"]
print("traffic sum GB", sum(b.values())/1e9)
print(" ".join("%s=%.2f" % (k, v/1e9) for k, v in sorted(b.items(), key=lambda kv: -kv[1])[:16]))
P
fi
