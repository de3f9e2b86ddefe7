#!/bin/bash
# decode attention: 8 instead of 4 timesteps per lane group and batch (KH_ATTN_UB, compile-time), experiment library
# kuiperllama_amd/lib/exp_ub8.so = kh_ops.hip + kh_model_step.hip rebuilt with -DKH_ATTN_UB=8 and linked with the other
# objects of kuiperllama_amd/build.py; same box, two alternating runs -> profiles/r4_attn_ub8.txt
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 300 python tools/attn_defer_ab.py llama3.2-1b 2>&1 | grep -v amdgpu | grep deferred | awk 'NR%2==1' | sed 's/^/base /'
  KH_LIB=$PWD/kuiperllama_amd/lib/exp_ub8.so timeout 300 python tools/attn_defer_ab.py llama3.2-1b 2>&1 | grep -v amdgpu | grep deferred | awk 'NR%2==1' | sed 's/^/ub8  /'
done > gpurun_out/r4_attn_ub8.txt
cat gpurun_out/r4_attn_ub8.txt | python3 -c "
import sys,json
rows={}
for l in sys.stdin:
    tag,js=l.split(' ',1); js=js.strip(); 
    if not js.startswith('{'): js=js[js.index('{'):]
    r=json.loads(js); rows.setdefault((r['pos']),{}).setdefault(tag,[]).append((r['attn_us'],r['step_us']))
for p in sorted(rows): print(p, {k:v for k,v in rows[p].items()})
"
