#!/bin/bash
# Round 6 experiment: one-barrier fold of the decode attention's lane groups (kh_attn.h, KH_ATTN_FOLD1=1;
# exp_fold1.so = python -m kuiperllama_amd.build --variant exp_fold1 KH_ATTN_FOLD1=1) vs the shipped two-barrier fold.
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
OUT=$O/r6_attn_fold_ab.txt
: > $OUT
L=$PWD/kuiperllama_amd/lib
# parity of the variant first: operator-level attention tests and the model tests that decode against the oracle
KH_LIB=$L/exp_fold1.so timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "mha or attn or golden or generate_modes or real_stride or full_size_baseline or prefill_is_bit or selftests" 2>&1 | tail -3 | tee -a $OUT
for i in $(seq 1 ${1:-3}); do
  for w in llama3.2-1b llama2-7b-int8; do
    python tools/kprof.py $w fold-2-barriers 2>&1 | tail -1 | tee -a $OUT
    KH_LIB=$L/exp_fold1.so python tools/kprof.py $w fold-1-barrier 2>&1 | tail -1 | tee -a $OUT
  done
done
for w in tinyllama-1.1b qwen2.5-0.5b; do
  python tools/kprof.py $w fold-2-barriers 2>&1 | tail -1 | tee -a $OUT
  KH_LIB=$L/exp_fold1.so python tools/kprof.py $w fold-1-barrier 2>&1 | tail -1 | tee -a $OUT
done
