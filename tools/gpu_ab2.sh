#!/bin/bash
# same-box A/B over several workloads: exp_base.so vs the working tree's library
R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { env $3 timeout 600 python bench.py --workload $2 --secondary "" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 $2', '%.1f tok/s' % d['value'], d['roofline']['kernels_avg_us'])"; }
{ for w in ${AB_WORKLOADS:-llama2-7b llama2-7b-int8 tinyllama-1.1b qwen2.5-0.5b}; do for i in 1 2; do run base $w KH_LIB=$R/kuiperllama_amd/lib/exp_base.so; run new $w KH_X=0; done; done; } > $O/r2_ab2.txt 2>&1
cat $O/r2_ab2.txt
