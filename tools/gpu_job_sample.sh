#!/bin/bash
# Round 6: k_sample requests *d_pos and the forced prompt token at kernel entry instead of behind the reduction
# (exp_prev.so = the library of the commit before: python -m kuiperllama_amd.build --variant-at <commit> exp_prev)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
OUT=$O/r6_sample_ab.txt
: > $OUT
L=$PWD/kuiperllama_amd/lib
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "generate or recycled or golden or stop or prompt or demo" 2>&1 | tail -3 | tee -a $OUT
for i in 1 2 3; do
  for w in llama3.2-1b llama2-7b-int8 stories15M; do
    KH_LIB=$L/exp_prev.so python tools/kprof.py $w sample-late-loads 2>&1 | tail -1 | tee -a $OUT
    python tools/kprof.py $w sample-early-loads 2>&1 | tail -1 | tee -a $OUT
  done
done
