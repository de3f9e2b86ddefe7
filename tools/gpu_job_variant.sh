#!/bin/bash
# Round 6: positions below 256 on a GQA model launch the per-head-only attention instantiation (step variant 2)
# instead of the one that also carries the group path (exp_prev.so = the library of the commit before).
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
OUT=$O/r6_attn_variant_ab.txt
: > $OUT
L=$PWD/kuiperllama_amd/lib
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "mha or attn or golden or generate_modes or crosses or selftests or token_parity or prefill_is_bit" 2>&1 | tail -2 | tee -a $OUT
for i in 1 2 3; do
  for w in llama3.2-1b llama2-7b-int8; do
    KH_LIB=$L/exp_prev.so python tools/kprof.py $w attn-with-group-code 2>&1 | tail -1 | tee -a $OUT
    python tools/kprof.py $w attn-per-head-only 2>&1 | tail -1 | tee -a $OUT
  done
done
for w in tinyllama-1.1b qwen2.5-0.5b; do
  KH_LIB=$L/exp_prev.so python tools/kprof.py $w attn-with-group-code 2>&1 | tail -1 | tee -a $OUT
  python tools/kprof.py $w attn-per-head-only 2>&1 | tail -1 | tee -a $OUT
done
