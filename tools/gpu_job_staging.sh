#!/bin/bash
# Round 6, VERDICT r5 item 1: the vector staging of the decode GEMVs.  Same-box, alternating A/B of
#   exp_base.so  the library of the round-5 HEAD (python -m kuiperllama_amd.build --variant-at 87acefa exp_base):
#                xs = w * (rs * x) staged behind a block sum (three barriers), 4 float4 slots per thread always
#   exp_mv4.so   the RMS scale applied in the epilogue (one barrier), still 4 slots per thread
#                (--variant exp_mv4 KH_STAGE_MAXV_MIN=4)
#   shipped      scale in the epilogue + the smallest staging depth that covers the vector (1 / 2 / 4 / 6)
# tools/kprof.py: tok/s over 128 greedy steps (best of 3) + back-to-back per-kernel us.  $1 = rounds (default 3).
# The full GPU suite runs first on the shipped library (parity gate).
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -x > $O/r6_pytest_staging.txt 2>&1
echo "pytest rc=$?" >> $O/r6_pytest_staging.txt
grep -E "passed|failed|FAILED|rc=" $O/r6_pytest_staging.txt | tail -8
OUT=$O/r6_staging_ab.txt
: > $OUT
L=$PWD/kuiperllama_amd/lib
for i in $(seq 1 ${1:-3}); do
  for w in llama3.2-1b llama2-7b-int8; do
    KH_LIB=$L/exp_base.so python tools/kprof.py $w base 2>&1 | tail -1 | tee -a $OUT
    KH_LIB=$L/exp_mv4.so python tools/kprof.py $w scale-in-epilogue 2>&1 | tail -1 | tee -a $OUT
    python tools/kprof.py $w scale-in-epilogue+exact-depth 2>&1 | tail -1 | tee -a $OUT
  done
done
for w in tinyllama-1.1b qwen2.5-0.5b; do
  KH_LIB=$L/exp_base.so python tools/kprof.py $w base 2>&1 | tail -1 | tee -a $OUT
  python tools/kprof.py $w scale-in-epilogue+exact-depth 2>&1 | tail -1 | tee -a $OUT
done
