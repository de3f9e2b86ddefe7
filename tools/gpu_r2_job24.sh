#!/bin/bash
# prompt attention: 8 waves + exp2-domain softmax; RCCL single-rank test
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_replicas_gloo.py -q -m gpu -x > $O/r2_rccl_test.log 2>&1; echo "rccl rc=$?"; grep -a "passed\|failed\|Error" $O/r2_rccl_test.log | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu -k "gemm or prefill" -x > $O/r2_gemm.log 2>&1; echo "rc=$?" >> $O/r2_gemm.log; tail -4 $O/r2_gemm.log
{ for w in llama3.2-1b llama2-7b-int8 qwen2.5-0.5b tinyllama-1.1b; do echo -n "== $w  "; timeout 300 python tools/pmc_workload.py $w --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill; done
  for pl in "1024 0" "128 1920" "128 8064"; do set -- $pl; echo -n "== llama3.2-1b prompt $1 pos0 $2  "; timeout 300 python tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 --prompt $1 --pos0 $2 2>&1 | grep prefill; done
  echo -n "== llama2-7b-int8 prompt 128 pos0 1920  "; timeout 300 python tools/pmc_workload.py llama2-7b-int8 --steps 0 --prefill gemm --reps 3 --prompt 128 --pos0 1920 2>&1 | grep prefill; } > $O/r2_attn8.txt 2>&1
cat $O/r2_attn8.txt
