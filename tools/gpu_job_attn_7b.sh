#!/bin/bash
# decode attention of Llama-2-7B int8 (32 MHA heads, head size 128): per-head split quantum 256 (KH_ATTN_TS=256, the
# r1-r4 constant) vs the per-geometry default (128 at head size 128, one split up to 256 timesteps), deferred and
# in-launch merge, positions inside and beyond the 128-step metric window.  Same box, alternating.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export AB_POS=63,127,255,256,383,511,767,1023,1535,2047
for rep in 1 2; do
for ts in 256 default; do
  if [ "$ts" = default ]; then unset KH_ATTN_TS; else export KH_ATTN_TS=$ts; fi
  timeout 400 python tools/attn_defer_ab.py llama2-7b-int8 2>&1 | grep -v amdgpu | grep '"model"' | awk 'NR<=20' | sed "s/^/ts$ts /"
done; done > gpurun_out/r5_attn_7b_quantum.txt
python3 - <<'PY'
import json
rows={}
for l in open('gpurun_out/r5_attn_7b_quantum.txt'):
    tag,js=l.split(' ',1)
    r=json.loads(js[js.index('{'):]); rows.setdefault((r['pos'],r['merge']),{}).setdefault(tag,[]).append((r['attn_us'],r['wo_us'],r['step_us']))
for k in sorted(rows): print(k, rows[k])
PY
