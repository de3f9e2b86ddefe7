#!/bin/bash
# round 3, GPU call B: ablation ladder of the int8 GEMV kernels, the rest of the GPU test suite
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 600 ./kuiperllama_amd/lib/mb_gemv_ladder > $O/r3_gemv_ladder.txt 2>&1
timeout 2000 python -m pytest tests -m gpu -q --timeout=900 > $O/r3_pytest_gpu_b.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_gpu_b.txt
cat $O/r3_gemv_ladder.txt; tail -25 $O/r3_pytest_gpu_b.txt
