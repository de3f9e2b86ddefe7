# to reproduce: git archive <round-1 commit> | tar -x -C _r1cmp ; (cd _r1cmp && python -m kuiperllama_amd.build)
#!/bin/bash
# same-box A/B of the round-1 tree against HEAD (decode tok/s and per-kernel us), alternating runs
R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { ( cd $1 && timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 ) | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary']
print('$2', 'fp32 %.1f tok/s' % d['value'], d['roofline']['kernels_avg_us'], '| int8 %.1f tok/s' % s['value'], s['roofline']['kernels_avg_us'])"; }
{ for i in 1 2; do run $R/_r1cmp r1; run $R head; done; } > $O/r2_r1_vs_head.txt 2>&1
cat $O/r2_r1_vs_head.txt
