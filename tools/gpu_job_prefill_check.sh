#!/bin/bash
# GEMM prefill: parity tests + speed of the four presets with / without the fused RoPE epilogue
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu -k "gemm or prefill" -x > $O/r2_gemm.log 2>&1; echo "rc=$?" >> $O/r2_gemm.log; tail -12 $O/r2_gemm.log
{ for w in llama3.2-1b llama2-7b-int8 qwen2.5-0.5b tinyllama-1.1b; do for e in 1 0; do echo -n "== $w KH_PG_ROPE_FUSE=$e  "; KH_PG_ROPE_FUSE=$e timeout 300 python tools/pmc_workload.py $w --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill; done; done; } > $O/r2_rope_fuse_ab.txt 2>&1
cat $O/r2_rope_fuse_ab.txt
