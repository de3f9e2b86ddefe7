#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O; R=$PWD
{
  for w in llama2-7b-int8 llama3.2-1b llama2-7b qwen2.5-0.5b tinyllama-1.1b; do
    KH_LIB=$R/kuiperllama_amd/lib/exp_r2.so timeout 300 python tools/kprof.py $w r2
    timeout 300 python tools/kprof.py $w r3
    KH_LIB=$R/kuiperllama_amd/lib/exp_r2.so timeout 300 python tools/kprof.py $w r2
    timeout 300 python tools/kprof.py $w r3
  done
} 2>&1 | grep -v amdgpu.ids > $O/r3_ab_auxfirst.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -k "not full_size" > $O/r3_pytest_gpu_g.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_gpu_g.txt
cat $O/r3_ab_auxfirst.txt; tail -4 $O/r3_pytest_gpu_g.txt
