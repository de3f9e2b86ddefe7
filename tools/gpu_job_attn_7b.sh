#!/bin/bash
# decode attention of Llama-2-7B int8 (32 MHA heads, head size 128): timesteps per split 256 (shipped) vs 128 / 64
# (experiment libraries built with -DKH_ATTN_MIN_TS), deferred and in-launch merge, positions inside and beyond the
# 128-step metric window.  Same box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export AB_POS=63,127,191,255,383,511,1023,2047
for lib in "" exp_ts128.so exp_ts64.so; do
  if [ -n "$lib" ]; then export KH_LIB=$PWD/kuiperllama_amd/lib/$lib; else unset KH_LIB; fi
  timeout 400 python tools/attn_defer_ab.py llama2-7b-int8 2>&1 | grep -v amdgpu | grep '"model"' | awk 'NR<=16' | sed "s/^/${lib:-base} /"
done > gpurun_out/r5_attn_7b_ts.txt
python3 - <<'PY'
import json
rows={}
for l in open('gpurun_out/r5_attn_7b_ts.txt'):
    tag,js=l.split(' ',1)
    r=json.loads(js[js.index('{'):]); rows.setdefault((r['pos'],r['merge']),{})[tag]=(r['attn_us'],r['wo_us'],r['step_us'])
for k in sorted(rows): print(k, rows[k])
PY
