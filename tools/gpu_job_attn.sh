#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; R=$PWD
{
  echo "== prev (last commit)"; KH_LIB=$R/kuiperllama_amd/lib/exp_prev.so KH_SWEEP_TLONGS=4096 timeout 400 python tools/attn_tlong_sweep.py llama3.2-1b 2>&1 | grep tlong
  echo "== new"; KH_SWEEP_TLONGS=4096 timeout 400 python tools/attn_tlong_sweep.py llama3.2-1b 2>&1 | grep tlong
  echo "== prev"; KH_LIB=$R/kuiperllama_amd/lib/exp_prev.so KH_SWEEP_TLONGS=4096 timeout 400 python tools/attn_tlong_sweep.py llama3.2-1b 2>&1 | grep tlong
  echo "== new"; KH_SWEEP_TLONGS=4096 timeout 400 python tools/attn_tlong_sweep.py llama3.2-1b 2>&1 | grep tlong
  KH_LIB=$R/kuiperllama_amd/lib/exp_prev.so timeout 300 python tools/kprof.py llama3.2-1b prev 2>&1 | grep label
  timeout 300 python tools/kprof.py llama3.2-1b new 2>&1 | grep label
} > $O/r3_attn_pipeline_ab.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --timeout=900 -k "mha or attention or real_stride or long_generate" > $O/r3_pytest_attn.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_attn.txt
cat $O/r3_attn_pipeline_ab.txt; tail -4 $O/r3_pytest_attn.txt
