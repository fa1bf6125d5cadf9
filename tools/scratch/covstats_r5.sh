#!/bin/bash
# per-kernel durations of the coverage call on config 3 (rocprofv3 kernel trace of tools/cov_prof.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/covstats; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 tools/cov_prof.py c3 64 5000000 10 > $O/cov_prof.txt 2> $O/err.txt
head -3 $O/cov_prof.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys,re
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    n=r['Name'].replace('(anonymous namespace)::','')
    if re.search(r'cb_|cl_|cov_|junc|jh_|scan|iv_', n):
        print('%-60s calls %5s avg %8.1f min %8.1f max %8.1f us' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
P
rm -rf $O/ks
