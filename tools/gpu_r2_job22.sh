#!/bin/bash
# the driver's multi-GPU launch form on the 1-GPU box: torchrun, world size 1, RCCL backend, config 5
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --workload llama2-7b --secondary "" ) > $O/r2_torchrun_ws1.json 2> $O/r2_torchrun_ws1.err; echo "rc=$?"
tail -3 $O/r2_torchrun_ws1.err; tail -1 $O/r2_torchrun_ws1.json | cut -c1-900
python - <<'P'
import json
d=json.loads(open("gpurun_out/r2_torchrun_ws1.json").read().strip().splitlines()[-1])
print(d["value"], d["config"], d.get("replicas"), d["cpu_baseline"].get("tokens_match_gpu"), d["cpu_baseline"].get("tokens_compared"))
P
