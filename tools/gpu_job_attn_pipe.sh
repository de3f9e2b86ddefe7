#!/bin/bash
# round 4: decode attention with two batches in flight (software-pipelined K/V loads) against the build before it
# (kuiperllama_amd/lib/exp_base.so), same box: parity tests first, then tools/attn_defer_ab.py on both libraries
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "attn or mha or split or long or defer or generate or prefill" > $O/r4_attn_pipe_tests.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r4_attn_pipe_tests.txt
tail -3 $O/r4_attn_pipe_tests.txt
{
  for rep in 1 2; do
    KH_LIB=$PWD/kuiperllama_amd/lib/exp_base.so timeout 400 python tools/attn_defer_ab.py ${AB_MODELS:-llama3.2-1b} 2>&1 | grep -v amdgpu | grep '^{' | sed 's/^/base /'
    timeout 400 python tools/attn_defer_ab.py ${AB_MODELS:-llama3.2-1b} 2>&1 | grep -v amdgpu | grep '^{' | sed 's/^/pipe /'
  done
} > $O/r4_attn_pipe_ab.txt 2>&1
python3 - <<'PY'
import json
rows={}
for l in open('gpurun_out/r4_attn_pipe_ab.txt'):
    tag,js=l.split(' ',1)
    try: r=json.loads(js)
    except Exception: continue
    rows.setdefault((r['model'],r['merge'],r['pos']),{}).setdefault(tag,[]).append((r['attn_us'],r['wo_us'],r['step_us']))
for k in sorted(rows):
    v=rows[k]
    f=lambda t: ' '.join('%.2f+%.2f|%.0f'%x for x in v.get(t,[]))
    print(k, 'base', f('base'), ' pipe', f('pipe'))
PY
