#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r3_prefill_ring28b.txt
: > $O
export KH_PT_SIZES=256,512
L=$PWD/kuiperllama_amd/lib
M="llama3.2-1b tinyllama-1.1b qwen2.5-0.5b llama2-7b"
timeout 600 python tools/prefill_time.py phases $M 2>/dev/null >> $O
KH_LIB=$L/exp_ring2.so timeout 600 python tools/prefill_time.py "ring D=2" $M 2>/dev/null >> $O
timeout 600 python tools/prefill_time.py phases-again $M 2>/dev/null >> $O
KH_LIB=$L/exp_ring2.so timeout 600 python tools/prefill_time.py "ring D=2 again" $M 2>/dev/null >> $O
cat $O
