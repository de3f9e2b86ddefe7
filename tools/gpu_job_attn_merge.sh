#!/bin/bash
# round 5: the split merge of the decode attention at 4 k ... 128 k positions (Llama-3.2-1B, GQA group path from
# position 4095 on): the r3-r4 two-pass merge (four dependent agent-scope trips at 32 splits) against the
# one-round-trip merge, the group path's split quantum 256 vs 128, and - as a TIMING PROBE ONLY - the publication
# without the storing waves' vmcnt(0) wait (what a protocol that does not serialise store -> ticket could gain).
# Same box, alternating; variant libraries built beforehand (python -m kuiperllama_amd.build --variant ...).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export AB_POS=${AB_POS:-2047,4094,4095,4096,6143,8191,16383,32768,65535,131071} AB_MODES=deferred
L=$PWD/kuiperllama_amd/lib
for rep in 1 2; do
for v in ${AB_VARIANTS:-exp_r5base shipped}; do
  if [ "$v" != shipped ]; then export KH_LIB=$L/$v.so; else unset KH_LIB; fi; tag=$v
  timeout 300 python tools/attn_defer_ab.py llama3.2-1b 2>&1 | grep -v amdgpu | grep '"model"' | sed "s/^/$tag /"
done; done > gpurun_out/${AB_OUT:-r5_attn_merge_ab2.txt}
python3 - <<'PY'
import json
rows={}
import os
for l in open('gpurun_out/' + os.environ.get('AB_OUT', 'r5_attn_merge_ab2.txt')):
    tag,js=l.split(' ',1)
    r=json.loads(js[js.index('{'):]); rows.setdefault(r['pos'],{}).setdefault(tag,[]).append((r['attn_us'],r['wo_us'],r['step_us']))
for k in sorted(rows):
    print(k, {t: v for t, v in rows[k].items()})
PY
