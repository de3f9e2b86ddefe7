#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O; R=$PWD
{
  timeout 300 python tools/prefill_time.py r3 llama3.2-1b
  for sh in "2,2,4" "2,4,4" "2,4,8" "1,4,8" "2,2,8" "1,4,2"; do
    KH_PG_SHAPE_RESID=$sh timeout 300 python tools/prefill_time.py "RESID=$sh" llama3.2-1b
  done
  for sh in "2,2,4" "1,4,4" "2,4,2" "2,4,8" "2,8,2"; do
    KH_PG_SHAPE_QKV=$sh timeout 300 python tools/prefill_time.py "QKV=$sh" llama3.2-1b
  done
  for sh in "2,8,1" "2,8,4" "2,4,2" "2,4,4"; do
    KH_PG_SHAPE_SWIGLU=$sh timeout 300 python tools/prefill_time.py "SWIGLU=$sh" llama3.2-1b
  done
} 2>&1 | grep -v amdgpu.ids > $O/r3_prefill_shapes.txt
cat $O/r3_prefill_shapes.txt
