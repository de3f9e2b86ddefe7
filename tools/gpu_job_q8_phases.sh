#!/bin/bash
# Round 6: the phase stamps of the Llama-2-7B int8 decode GEMVs (tools/mb_q8ring.hip built with -DKH_TRACE) on the
# round-6 staging (g = w * x staged at once, RMS scale in the epilogue) - the "after" to profiles/r5_q8_phases.txt -
# and the untraced microbenchmark beside it (bitwise comparison ring vs register tiles included).
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
OUT=$O/r6_q8_phases.txt
: > $OUT
for i in 1 2; do
  echo "== mb_q8ring (untraced), all five kernels, 4 sweeps per graph" >> $OUT
  timeout 300 kuiperllama_amd/lib/mb_q8ring -1 0 4 2>&1 | grep -v amdgpu.ids >> $OUT
  echo "== mb_q8ring built with -DKH_TRACE" >> $OUT
  timeout 300 kuiperllama_amd/lib/mb_q8ring_trace -1 0 1 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT | cut -c1-330
