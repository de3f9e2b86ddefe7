#!/bin/bash
# round 4: the split merge of the decode attention as a launch of its own (k_attn_merge, step variants 3 / 4) against
# the forms it replaces (KH_ATTN_MERGE_LAUNCH=0: k_wo_comb up to 16 splits, last-arriver merge on the group path), same
# library, same box: parity tests first, then tools/attn_defer_ab.py both ways, interleaved
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "attn or mha or split or long or defer or generate" > $O/r4_attn_merge_tests.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r4_attn_merge_tests.txt
tail -3 $O/r4_attn_merge_tests.txt
{
  for rep in 1 2; do
    KH_ATTN_MERGE_LAUNCH=0 timeout 400 python tools/attn_defer_ab.py ${AB_MODELS:-llama3.2-1b} 2>&1 | grep -v amdgpu | grep '^{' | grep deferred | sed 's/^/base /'
    timeout 400 python tools/attn_defer_ab.py ${AB_MODELS:-llama3.2-1b} 2>&1 | grep -v amdgpu | grep '^{' | grep deferred | sed 's/^/new  /'
  done
} > $O/r4_attn_merge_ab.txt 2>&1
python3 - <<'PY'
import json
rows={}
for l in open('gpurun_out/r4_attn_merge_ab.txt'):
    tag,js=l.split(None,1)
    try: r=json.loads(js)
    except Exception: continue
    rows.setdefault((r['model'],r['pos']),{}).setdefault(tag,[]).append((r['attn_us'],r['wo_us'],r['step_us']))
for k in sorted(rows):
    v=rows[k]
    f=lambda t: ' '.join('%5.2f+%4.2f|%4.0f'%x for x in v.get(t,[]))
    print('%-13s %6d  base %s   new %s'%(k[0],k[1],f('base'),f('new')))
PY
