#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B=kuiperllama_amd/lib/mb_grid_barrier
O=gpurun_out/r3_launch_env.txt
: > $O
for e in "" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" "ROC_USE_FGS_KERNARG=0" "AMD_DIRECT_DISPATCH=0"; do
  echo "== ${e:-default}" >> $O
  env $e timeout 60 $B 2>&1 | grep -E "A graph|E nosync" | tail -2 >> $O
done
cat $O
