#!/bin/bash
# Llama-3.2-1B fp32 ffn13 / w2 on the LDS-DMA ring core vs the shipped register-tile kernels
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
F=$O/r6_fp32_ring_ab.txt
( timeout 600 kuiperllama_amd/lib/mb_f32ring -1 2; echo "--- second pass, 4 sweeps per graph"; timeout 600 kuiperllama_amd/lib/mb_f32ring -1 4 ) > $F 2>&1
cat $F
