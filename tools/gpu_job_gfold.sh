#!/bin/bash
# Round 6: one-barrier fold on the GQA group path of the decode attention (long contexts) against the two-barrier fold
# (exp_prev.so = the library of the commit before, --variant-at); parity tests of the deep positions first.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "mha or attn or real_stride or long or deep or selftest or group" 2>&1 | tail -3 > gpurun_out/${GF_OUT:-r6_attn_gfold_ab.txt}
AB_OUT=${GF_OUT:-r6_attn_gfold}_raw.txt AB_VARIANTS="exp_prev shipped" AB_POS=4094,4095,4096,8191,16383,32768,65535,131071 bash tools/gpu_job_attn_merge.sh >> gpurun_out/${GF_OUT:-r6_attn_gfold_ab.txt} 2>&1
cat gpurun_out/${GF_OUT:-r6_attn_gfold_ab.txt}
