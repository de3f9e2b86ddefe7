#!/bin/bash
# MFMA prefill attention: parity tests, then speed A/B against the decode-kernel attention
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "gemm or prefill_api" -x > $O/r2_gemm.log 2>&1; echo "gemm rc=$?" >> $O/r2_gemm.log
tail -30 $O/r2_gemm.log
{
for w in llama3.2-1b llama2-7b-int8 qwen2.5-0.5b tinyllama-1.1b; do for e in 1 0; do echo -n "== $w KH_PG_ATTN=$e  "; KH_PG_ATTN=$e timeout 300 python tools/pmc_workload.py $w --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill; done; done
for e in 1 0; do for pl in "1024 0" "128 1920" "128 8064"; do set -- $pl; echo -n "== llama3.2-1b KH_PG_ATTN=$e prompt $1 pos0 $2  "; KH_PG_ATTN=$e timeout 300 python tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 --prompt $1 --pos0 $2 2>&1 | grep prefill; done; done
} > $O/r2_attn_mfma_ab.txt 2>&1
cat $O/r2_attn_mfma_ab.txt
