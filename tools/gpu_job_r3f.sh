#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests -m gpu -q -k "prefill" --timeout=900 > $O/r3_pytest_prefill.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_prefill.txt
{
  for i in 1 2; do
    KH_LIB=$R/kuiperllama_amd/lib/exp_r2.so timeout 300 python tools/prefill_time.py r2 llama3.2-1b tinyllama-1.1b qwen2.5-0.5b llama2-7b-int8
    timeout 300 python tools/prefill_time.py r3 llama3.2-1b tinyllama-1.1b qwen2.5-0.5b llama2-7b-int8
  done
} 2>&1 | grep -v amdgpu.ids > $O/r3_prefill_ab2.txt
tail -30 $O/r3_pytest_prefill.txt; cat $O/r3_prefill_ab2.txt
