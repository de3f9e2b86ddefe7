#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu -k "gemm or prefill" -x > $O/r2_gemm.log 2>&1; echo "rc=$?" >> $O/r2_gemm.log; tail -3 $O/r2_gemm.log
{ for pl in "128 0" "128 1920"; do set -- $pl; echo -n "== llama2-7b-int8 prompt $1 pos0 $2  "; timeout 300 python tools/pmc_workload.py llama2-7b-int8 --steps 0 --prefill gemm --reps 3 --prompt $1 --pos0 $2 2>&1 | grep prefill; done; } > $O/r2_attn8_hs128.txt 2>&1
cat $O/r2_attn8_hs128.txt
