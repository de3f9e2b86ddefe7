#!/bin/bash
# same-box A/B: kuiperllama_amd/lib/exp_base.so (built from the committed sources) vs the working tree's library, alternating
R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { env $2 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary']
print('$1', 'fp32 %.1f' % d['value'], d['roofline']['kernels_avg_us'], '| int8 %.1f' % s['value'], s['roofline']['kernels_avg_us'])"; }
{ for i in 1 2 3; do run base KH_LIB=$R/kuiperllama_amd/lib/exp_base.so; run new ${AB_NEW:-KH_X=0}; done; } > $O/r2_ab.txt 2>&1
cat $O/r2_ab.txt
