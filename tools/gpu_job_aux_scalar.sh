#!/bin/bash
# Round 6 experiment: the epilogue operands of the register-tile GEMVs (sin / cos / bias in k_qkv, the residual pair in
# k_gemv_res) fetched through the scalar cache (s_load, lgkmcnt) instead of wave-uniform vector loads - two VMEM
# instructions fewer per work item on the CU's address path.  exp_auxs.so = --variant exp_auxs KH_AUX_SCALAR=1 on the
# experiment tree (ld_uniform in kh_fused.h's PRE lambdas behind that macro; measured: no gain, macro removed -
# profiles/r6_aux_scalar_ab.txt).
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
OUT=$O/r6_aux_scalar_ab.txt
: > $OUT
L=$PWD/kuiperllama_amd/lib
for i in $(seq 1 ${1:-3}); do
  for w in llama3.2-1b llama2-7b-int8; do
    python tools/kprof.py $w vector-aux 2>&1 | tail -1 | tee -a $OUT
    KH_LIB=$L/exp_auxs.so python tools/kprof.py $w scalar-aux 2>&1 | tail -1 | tee -a $OUT
  done
done
for w in tinyllama-1.1b qwen2.5-0.5b; do
  python tools/kprof.py $w vector-aux 2>&1 | tail -1 | tee -a $OUT
  KH_LIB=$L/exp_auxs.so python tools/kprof.py $w scalar-aux 2>&1 | tail -1 | tee -a $OUT
done
