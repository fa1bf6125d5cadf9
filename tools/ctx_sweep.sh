cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ctx
for c in 4 3 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-path --no-e2e --steps 24 --warmup 6 --contexts $c > gpurun_out/ctx/c$c.json 2> gpurun_out/ctx/c$c.err || { tail -3 gpurun_out/ctx/c$c.err; exit 1; }
  python - <<P
import json
d=json.load(open("gpurun_out/ctx/c$c.json"))
print("contexts $c ms_per_step", d["ms_per_step"], "hbm", d["config"]["hbm_in_use_gb"])
P
done
