#!/bin/bash
# SQ counters of the default bench's kernels (one PMC pass): where the wave cycles of a kernel go.
# usage (through gpurun): bash tools/pmc_sq.sh <tag> [kernel-name-substring]
tag=${1:-pmc}; pat=${2:-wg_hash}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${PMC_OUT:-gpurun_out}/$tag; mkdir -p $O
rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS} --output-format csv -d $O/p -o p -- python3 bench.py --steps 2 --warmup 1 --prof-steps 1 --no-cpu-baseline --no-host-path --no-e2e > $O/bench.json 2> $O/err.log
f=$(find $O/p -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    import re
    m = re.search(r"(\w+_k)\b", k)
    k = (m.group(1) if m else k[:60]) + ("<raw>" if "ILb1" in r["Kernel_Name"] or "<true>" in r["Kernel_Name"] else "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    n = max(cnt[(k, c)] for c in d)
    print(k, "launches", n)
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    for c, v in sorted(d.items()):
        print("   %-22s %14.0f per launch   %.3f of WAVE_CYCLES" % (c, v / n, v / wc))
P
rm -rf $O/p
