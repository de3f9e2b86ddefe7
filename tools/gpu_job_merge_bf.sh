#!/bin/bash
# Round 6: the split merges without their per-split (uniform) branches - attn_merge_lds (last arriver, group path) and
# CombStager (k_wo_comb, positions 256 ... 4094): a slot past the active count adds an exact zero, and the LDS reads leave
# together.  exp_prev = the library of the commit before (--variant-at); parity tests of the split paths first.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F=gpurun_out/r6_attn_merge_bf_ab.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "mha or attn or real_stride or crosses or deferred or selftest or stress" 2>&1 | tail -3 > $F
AB_OUT=r6_attn_merge_bf_raw.txt AB_VARIANTS="exp_prev shipped" AB_POS=255,256,511,1023,2047,4094,4095,4096,8191,16383,32768,131071 bash tools/gpu_job_attn_merge.sh >> $F 2>&1
cat $F
