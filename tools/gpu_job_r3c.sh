#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O; R=$PWD
{
  for w in llama2-7b-int8 llama3.2-1b llama2-7b qwen2.5-0.5b tinyllama-1.1b; do
    KH_LIB=$R/kuiperllama_amd/lib/exp_r2.so timeout 300 python tools/kprof.py $w r2
    KH_SHAPE_DEBUG=1 timeout 300 python tools/kprof.py $w r3
  done
} > $O/r3_ab_shapes.txt 2>&1
timeout 900 python tools/attn_tlong_sweep.py llama3.2-1b > $O/r3_attn_tlong.txt 2>&1
timeout 600 python tools/attn_tlong_sweep.py qwen2.5-0.5b >> $O/r3_attn_tlong.txt 2>&1
timeout 2000 python -m pytest tests -m gpu -q --timeout=900 > $O/r3_pytest_gpu_c.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_gpu_c.txt
grep -v amdgpu.ids $O/r3_ab_shapes.txt; grep tlong $O/r3_attn_tlong.txt; tail -6 $O/r3_pytest_gpu_c.txt
