#!/bin/bash
mkdir -p gpurun_out/n2
TBK_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/n2/bench.json 2> gpurun_out/n2/bench.err; echo rc=$?
tail -3 gpurun_out/n2/bench.err | cut -c1-300
python - <<P
import json
l=[x for x in open("gpurun_out/n2/bench.json").read().strip().splitlines() if x.startswith("{")]
d=json.loads(l[-1])
print({k:d[k] for k in ("value","n_gpus","ms_per_step","scaling","plain_ms_per_step","wire_bytes_per_step","partials_per_step","cut_rounds_max")})
print(d["config"])
P
