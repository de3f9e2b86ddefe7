#!/bin/bash
# Round 6: the reader of the one-barrier attention fold - eight exec-masked LDS round trips (the form that went in
# first, exp_prev.so = the library of the commit before) against the branch-free reader, and round 5 (tools/_r5tree)
# on the same box, alternating.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
OUT=$O/r6_attn_fold_reader_ab.txt
: > $OUT
L=$PWD/kuiperllama_amd/lib
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "mha or attn or golden or generate_modes or real_stride or crosses or selftests or token_parity" 2>&1 | tail -2 | tee -a $OUT
for i in 1 2 3; do
  for w in llama3.2-1b llama2-7b-int8; do
    ( cd tools/_r5tree && python tools/kprof.py $w round5 2>&1 | tail -1 ) | tee -a $OUT
    KH_LIB=$L/exp_prev.so python tools/kprof.py $w fold-masked-reader 2>&1 | tail -1 | tee -a $OUT
    python tools/kprof.py $w fold-branch-free-reader 2>&1 | tail -1 | tee -a $OUT
  done
done
for w in tinyllama-1.1b qwen2.5-0.5b; do
  ( cd tools/_r5tree && python tools/kprof.py $w round5 2>&1 | tail -1 ) | tee -a $OUT
  KH_LIB=$L/exp_prev.so python tools/kprof.py $w fold-masked-reader 2>&1 | tail -1 | tee -a $OUT
  python tools/kprof.py $w fold-branch-free-reader 2>&1 | tail -1 | tee -a $OUT
done
