#!/bin/bash
# usage: tools/run_variants.sh <workload> : bench the shipped lib and every lib/exp_*.so
cd "$(dirname "$0")/.."
W=${1:-llama2-7b-int8}
for lib in kuiperllama_amd/lib/libkuiper_hip.so kuiperllama_amd/lib/exp_*.so; do
  KH_LIB=$PWD/$lib timeout 300 python bench.py --workload $W --secondary "" --no-cpu-baseline --repeats 1 --steps 128 --warmup 16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], '$W', round(d['value'],1), d['roofline']['kernels_avg_us'])"
done
