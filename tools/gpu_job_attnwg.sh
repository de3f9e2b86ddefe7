#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_attn_wg.txt; : > $O
for w in llama2-7b-int8 llama3.2-1b; do
  for wg in 512 256; do
    KH_ATTN_WG=$wg timeout 300 python tools/kprof.py $w "attn_wg=$wg" 2>/dev/null >> $O
  done
done
cat $O
