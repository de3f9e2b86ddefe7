#!/bin/bash
# decode attention per layer at deep positions: splits per KV group on the GQA group path (experiment libraries
# built with -DKH_ATTN_MAX_NS_G=<n> for every .hip of kuiperllama_amd/build.py, KH_LIB selects them)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r3_attn_nsg.txt
: > $O
L=$PWD/kuiperllama_amd/lib
export KH_SWEEP_TLONGS=4096
for v in shipped exp_nsg16 exp_nsg24 exp_nsg48 exp_nsg64; do
  echo "== $v" >> $O
  if [ $v = shipped ]; then timeout 600 python tools/attn_tlong_sweep.py 2>/dev/null >> $O
  else KH_LIB=$L/$v.so timeout 600 python tools/attn_tlong_sweep.py 2>/dev/null >> $O; fi
done
cat $O
