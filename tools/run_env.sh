#!/bin/bash
# usage: tools/run_env.sh <workload> "<ENV=VAL ...>" ... : bench under each env set
cd "$(dirname "$0")/.."
W=$1; shift
for e in "$@"; do
  env $e timeout 300 python bench.py --workload $W --secondary "" --no-cpu-baseline --steps 128 --warmup 16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$e]', round(d['value'],1), d['roofline']['kernels_avg_us'])"
done
