#!/bin/bash
# Round 6's later additions to the evidence under profiles/ (the command line after ABI 8): through gpurun, after tools/full_check.sh.
# Leaves its files under gpurun_out/prof/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
# the command line on 32 x 1 M reads with SEQ / QUAL: phase lines, then kernels and copies of one run under rocprofv3
rm -rf gpurun_out/e2e_prof
bash tools/e2e_prof.sh 32 1000000 > $O/e2e_seq_cli.txt 2>&1
cp gpurun_out/e2e_prof/e2e_kernel_stats.csv $O/e2e_seq_kernel_stats.csv
cp gpurun_out/e2e_prof/e2e_memory_copy_stats.csv $O/e2e_seq_memory_copy_stats.csv
rm -rf gpurun_out/e2e_prof /tmp/tbk_e2e_prof
# the end-to-end legs as bench.py runs them (child processes), alone, with the CPU files -> files baseline
python3 tools/e2e_leg.py --cpu-baseline > $O/e2e_legs.json 2> $O/e2e_legs.err || true
# a run and its predecessor: the bare-records leg six times back to back, then six times a second apart (phase lines of the library)
{ echo "== back to back"; STALL_PAUSE=0 python3 tools/stall_probe.py 32 1000000 6 0 TBK_DEBUG=phases=1; echo "== one second apart"; STALL_PAUSE=1.0 python3 tools/stall_probe.py 32 1000000 6 0 TBK_DEBUG=phases=1; } 2>&1 | grep -v amdgpu.ids | grep -v "1 groups" > $O/e2e_stall_probe.txt
# what a first copy into a freshly page-locked block costs (tools/micro/d2h_first.hip)
hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_first tools/micro/d2h_first.hip 2> /dev/null && { /tmp/d2h_first 20; /tmp/d2h_first 120; } > $O/d2h_first.txt 2>&1
echo refresh d done; ls -la $O
