#!/bin/bash
# forced GEMM shapes for a 128- and a 512-token pass on Llama-3.2-1B (KH_PG_SHAPE_<EPI>="R,NT,ks[,kz]")
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r3_prefill_shapes512.txt
: > $O
export KH_PT_SIZES=512
timeout 300 python tools/prefill_time.py auto llama3.2-1b 2>/dev/null >> $O
for v in "QKV=2,8,4" "QKV=2,4,2" "QKV=2,4,8" "QKV=1,4,4" "SWIGLU=2,8,4" "SWIGLU=2,8,1" "SWIGLU=2,4,2" "SWIGLU=2,4,4" "RESID=2,8,8" "RESID=2,8,8,2" "RESID=2,8,4,2" "RESID=2,4,4" "RESID=2,4,8,2"; do
  env KH_PG_SHAPE_$v timeout 300 python tools/prefill_time.py "$v" llama3.2-1b 2>/dev/null >> $O
done
cat $O
