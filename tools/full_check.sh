#!/bin/bash
# the whole GPU suite, then the default bench line with its end-to-end legs (through gpurun): bash tools/full_check.sh [tag]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-full}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<P
import json
d=json.load(open("$O/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "h2h", d.get("value_host_to_host"), "link", d.get("roofline_link",{}).get("frac"))
for k in ("end_to_end","end_to_end_seq","end_to_end_seq_long","end_to_end_c3_options"):
    v=d[k]; print(k, v["value"], v["wall_s_min"], v["wall_s"], v["wall_s_max"])
print("cpu e2e", d["cpu_baseline"].get("end_to_end",{}).get("value"))
P
