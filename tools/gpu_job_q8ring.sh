#!/bin/bash
# Same-box check of the LDS-DMA ring kernels inside the model: the Llama-2-7B int8 step with the ring kernels off / on,
# alternating (tools/kprof.py: tok/s over 128 steps + back-to-back per-kernel us).  $1 = rounds (default 3).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r5_q8ring_model.txt
: > $OUT
for i in $(seq 1 ${1:-3}); do
  KH_RING=0 python tools/kprof.py llama2-7b-int8 register-tiles 2>&1 | tail -1 | tee -a $OUT
  python tools/kprof.py llama2-7b-int8 lds-dma-ring 2>&1 | tail -1 | tee -a $OUT
done
python tools/kprof.py llama3.2-1b fp32 2>&1 | tail -1 | tee -a $OUT
