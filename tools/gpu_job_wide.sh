#!/bin/bash
# prompt-prefill record of the round: prompt tok/s by prompt length (shipped, 128-token passes, no solo
# launches, no K split across workgroups) and the per-kernel table of a 128- and a 512-token pass
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r3_prefill_wide.txt
: > $O
M="llama3.2-1b llama2-7b-int8 tinyllama-1.1b qwen2.5-0.5b llama2-7b"
timeout 900 python tools/prefill_time.py shipped $M 2>/dev/null >> $O
env KH_PG_CHUNK=128 timeout 900 python tools/prefill_time.py KH_PG_CHUNK=128 $M 2>/dev/null >> $O
env KH_PG_SOLO=0 timeout 600 python tools/prefill_time.py "KH_PG_SOLO=0" llama3.2-1b qwen2.5-0.5b llama2-7b 2>/dev/null >> $O
env KH_PG_KZ=0 KH_PT_SIZES=256 timeout 600 python tools/prefill_time.py "KH_PG_KZ=0" $M 2>/dev/null >> $O
cat $O
for w in llama3.2-1b llama2-7b-int8; do
  rm -rf /tmp/pfk_$w
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pfk_$w -- python tools/prefill_kernels.py run $w 128 512 > gpurun_out/r3_pfk_$w.log 2>&1
  f=$(find /tmp/pfk_$w -name "*kernel_trace.csv" | head -1)
  { grep "prompt tok/s" gpurun_out/r3_pfk_$w.log; python tools/prefill_kernels.py table "$f"; } | tee gpurun_out/r3_prefill_wide_kernels_$w.txt
done
