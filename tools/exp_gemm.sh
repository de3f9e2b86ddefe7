#!/bin/bash
# Ablation variants of the GEMM prefill (WRONG results on purpose): which operand stream limits it?
set -e
cd "$(dirname "$0")/.."
L=kuiperllama_amd/lib; C=kuiperllama_amd/csrc
for v in NOA:-DKH_PG_EXP_NOA=1 NOB:-DKH_PG_EXP_NOB=1 NOAB:"-DKH_PG_EXP_NOA=1 -DKH_PG_EXP_NOB=1"; do
  name=${v%%:*}; mac=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared $mac $C/kh_ops.hip $C/kh_model.hip $C/kh_tokenizer.cpp $C/kh_bpe.cpp -o $L/exp_pg_${name}.so &
done
wait
ls -la $L/exp_pg_*.so
