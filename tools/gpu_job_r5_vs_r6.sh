#!/bin/bash
# Round 6 against round 5 on ONE box, alternating: tools/_r5tree = `git archive 87acefa kuiperllama_amd include
# tools/kprof.py` with its own library built in place (untracked scratch copy, see the header of
# profiles/r6_r5_vs_r6_same_box.txt), against the tree as it is.  tools/kprof.py: 128 greedy steps, hipGraph, best of 3.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
OUT=$O/r6_r5_vs_r6_same_box.txt
: > $OUT
for i in 1 2 3; do
  for w in llama3.2-1b llama2-7b-int8; do
    ( cd tools/_r5tree && python tools/kprof.py $w round5 2>&1 | tail -1 ) | tee -a $OUT
    python tools/kprof.py $w round6 2>&1 | tail -1 | tee -a $OUT
  done
done
for w in tinyllama-1.1b qwen2.5-0.5b llama2-7b; do
  ( cd tools/_r5tree && python tools/kprof.py $w round5 2>&1 | tail -1 ) | tee -a $OUT
  python tools/kprof.py $w round6 2>&1 | tail -1 | tee -a $OUT
done
