#!/bin/bash
# round-2 GPU job 6: wo-weight prefetch riding on the decode attention launch: A/B + parity
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
tools/run_env.sh llama3.2-1b "KH_ATTN_PF=0" "KH_ATTN_PF=1"
tools/run_env.sh llama2-7b-int8 "KH_ATTN_PF=0" "KH_ATTN_PF=1"
tools/run_env.sh tinyllama-1.1b "KH_ATTN_PF=0" "KH_ATTN_PF=1"
} > $O/r2_attn_pf_ab.txt 2>&1
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "full_size_baseline or token_parity or long_generate or generate_modes" > $O/r2_pf_parity.log 2>&1; echo "rc=$?" >> $O/r2_pf_parity.log
cat $O/r2_attn_pf_ab.txt; tail -4 $O/r2_pf_parity.log
