#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "mha_prefill" -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "gemm_prefill or long_prompt" -x 2>&1 | tail -3
O=gpurun_out/r3_pattn_qt.txt
: > $O
for cfg in "" "KH_PG_ATTN_QT=1" "KH_PG_ATTN_QT=2" "KH_PG_ATTN_QT=4"; do
  env $cfg timeout 600 python tools/pattn_time.py "${cfg:-auto}" llama3.2-1b 2>/dev/null >> $O
done
for cfg in "" "KH_PG_ATTN_QT=1"; do
  env $cfg timeout 600 python tools/pattn_time.py "${cfg:-auto}" llama2-7b-int8 qwen2.5-0.5b 2>/dev/null >> $O
done
cat $O
