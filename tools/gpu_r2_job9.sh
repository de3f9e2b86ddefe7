#!/bin/bash
# GEMM prefill shape sweep (Llama-3.2-1B, 128 tokens): ms per prefill under forced shapes
R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { echo -n "[$*] "; env "$@" timeout 120 python tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill; }
{
run KH_X=0
for s in 2,8,4 2,8,2 2,4,4 2,4,2 2,4,1 2,2,2 2,2,4 1,4,4; do run KH_PG_SHAPE_SWIGLU=$s; done
for s in 2,8,8 2,4,8 2,4,4 2,2,8 2,2,4 2,2,2 1,4,8 1,4,4; do run KH_PG_SHAPE_RESID=$s; done
for s in 2,8,8 2,4,8 2,4,4 2,2,8 2,2,4 1,4,4 1,4,8; do run KH_PG_SHAPE_QKV=$s; done
} > $O/r2_gemm_shape_sweep.txt 2>&1
cat $O/r2_gemm_shape_sweep.txt
