#!/bin/bash
# round 4: the persistent decode-layer engine prototype against the five-launch chain (tools/mb_engine.hip)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
B=${ENGINE_BIN:-kuiperllama_amd/lib/mb_engine}
{
  for cfg in ${ENGINE_CFGS:-1b qwen tiny}; do
    for pos in ${ENGINE_POS:-64}; do
      echo "== $cfg pos $pos"; timeout 120 $B $cfg $pos 20; echo "rc=$?"
    done
  done
} > $O/r4_engine.txt 2>&1
cat $O/r4_engine.txt
