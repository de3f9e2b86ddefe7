#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O; R=$PWD
{
  for w in llama2-7b-int8 llama3.2-1b qwen2.5-0.5b llama2-7b; do
    for rep in 1 2; do
      for l in exp_r2.so exp_bulk.so exp_mixed.so ""; do
        if [ -n "$l" ]; then KH_LIB=$R/kuiperllama_amd/lib/$l timeout 300 python tools/kprof.py $w ${l%.so}; else timeout 300 python tools/kprof.py $w roll; fi
      done
    done
  done
} 2>&1 | grep -v amdgpu.ids > $O/r3_ab_roll_bulk.txt
cat $O/r3_ab_roll_bulk.txt
