#!/bin/bash
# decode attention: minimum timesteps per time split (KH_ATTN_MIN_TS, compile-time) 256 (shipped) vs 128 / 64, experiment
# libraries kuiperllama_amd/lib/exp_ts*.so (all .hip units rebuilt with -DKH_ATTN_MIN_TS=<n>), same box
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for lib in "" exp_ts128.so exp_ts64.so; do
    if [ -n "$lib" ]; then export KH_LIB=$PWD/kuiperllama_amd/lib/$lib; else unset KH_LIB; fi
    timeout 300 python tools/attn_defer_ab.py llama3.2-1b 2>&1 | grep -v amdgpu | grep '"deferred"' | awk 'NR<=14' | sed "s/^/${lib:-base} /"
  done
done > gpurun_out/r4_attn_min_ts.txt
python3 - <<'PY'
import json
rows={}
for l in open('gpurun_out/r4_attn_min_ts.txt'):
    tag,js=l.split(' ',1)
    r=json.loads(js[js.index('{'):]); rows.setdefault(r['pos'],{}).setdefault(tag,[]).append((r['attn_us'],r['wo_us'],r['step_us']))
for p in sorted(rows): print(p, {k:v for k,v in rows[p].items()})
PY
