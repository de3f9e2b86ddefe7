#!/bin/bash
# ffn13 / cls, shipped vs ring: weight arrays on / 32 B off a 4-KiB boundary, graphs of 32 and 256 launches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r5_q8ring_4.txt
: > $OUT
for k in 0 1; do for s in 0 32; do for r in 1 8; do
  echo "== kernel $k (0 ffn13, 1 cls)  shift $s B  sweeps per graph $r" >> $OUT
  timeout 120 kuiperllama_amd/lib/mb_q8ring $k $s $r 2>&1 | sed -n 2,5p >> $OUT
done; done; done
cat $OUT
