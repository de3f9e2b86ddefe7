#!/bin/bash
# same-box A/B: kuiperllama_amd/lib/exp_prev.so (built from the last commit) vs the working tree's library,
# alternating, per-kernel back-to-back times + tok/s (tools/kprof.py).  usage: tools/gpu_job_ab.sh out.txt workload...
cd "$(dirname "$0")/.."
O=gpurun_out/$1; shift; R=$PWD
{
  for w in "$@"; do
    for rep in 1 2; do
      KH_LIB=$R/kuiperllama_amd/lib/exp_prev.so timeout 300 python tools/kprof.py $w prev
      timeout 300 python tools/kprof.py $w new
    done
  done
} 2>&1 | grep -v amdgpu.ids > $O
python - "$O" <<'PY'
import sys, json
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); k = d['kernels_us']
        print('%-6s %-15s %7.1f  qkv %.2f attn %.2f wo %.2f ffn %.2f w2 %.2f cls %.2f' % (d['label'], d['workload'], d['tok_s'], k['qkv'], k['attn'], k['wo'], k['ffn13'], k['w2'], k['cls']))
    else:
        print(ln.rstrip())
PY
