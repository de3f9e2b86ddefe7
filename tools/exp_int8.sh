#!/bin/bash
# Build experiment variants of the library (ablation of the int8 GEMV inner loop) next to the
# shipped one.  Variants compute WRONG results on purpose; they only bound the cost of one
# ingredient.  Run with KH_LIB=<path> python bench.py ...
set -e
cd "$(dirname "$0")/.."
L=kuiperllama_amd/lib; C=kuiperllama_amd/csrc
for v in NOSTAGE:KH_EXP_NOSTAGE NOSCALE:KH_EXP_NOSCALE NOLDS:KH_EXP_NOLDS NOCVT:KH_EXP_NOCVT; do
  name=${v%%:*}; mac=${v##*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -D${mac}=1 $C/kh_ops.hip $C/kh_model.hip -o $L/exp_${name}.so &
done
wait
ls -la $L/exp_*.so
