#!/bin/bash
# round-2 GPU job 4: GEMM prefill ablations (which operand stream limits the MFMA rate?)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for lib in libkuiper_hip.so exp_pg_NOA.so exp_pg_NOB.so exp_pg_NOAB.so; do
  echo "== $lib"; KH_LIB=$R/kuiperllama_amd/lib/$lib timeout 200 python tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill
done > $O/r2_gemm_ablation.txt 2>&1
cat $O/r2_gemm_ablation.txt
