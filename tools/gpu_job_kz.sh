#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "prefill or long_prompt" -x 2>&1 | tail -3
O=gpurun_out/r3_prefill_kz.txt
: > $O
export KH_PT_SIZES=256,512
M="llama3.2-1b tinyllama-1.1b qwen2.5-0.5b llama2-7b-int8 llama2-7b"
timeout 900 python tools/prefill_time.py auto $M 2>/dev/null >> $O
KH_PG_KZ=0 timeout 900 python tools/prefill_time.py KH_PG_KZ=0 $M 2>/dev/null >> $O
export KH_PT_SIZES=
for sh in "2,8,4,4" "2,8,4,2" "2,4,4,2" "2,4,4,4" "1,4,4,2"; do
  KH_PG_SHAPE_RESID=$sh timeout 600 python tools/prefill_time.py "RESID=$sh" llama3.2-1b tinyllama-1.1b 2>/dev/null >> $O
done
for sh in "2,8,8,2" "2,8,4,2" "2,8,4,4" "2,4,8,2"; do
  KH_PG_SHAPE_RESID=$sh timeout 600 python tools/prefill_time.py "RESID=$sh" llama2-7b-int8 2>/dev/null >> $O
done
cat $O
