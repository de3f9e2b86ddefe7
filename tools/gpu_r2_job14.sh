#!/bin/bash
# op-level MFMA attention test + per-kernel GEMM ablation (which operand stream limits the small-M GEMMs)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "mha_prefill" -x > $O/r2_pattn.log 2>&1; echo "pattn rc=$?" >> $O/r2_pattn.log
tail -15 $O/r2_pattn.log
export TMPDIR=/tmp; cd /tmp
{
for v in default NOA NOB NOAB; do
  lib=$R/kuiperllama_amd/lib/exp_pg_$v.so; [ $v = default ] && lib=$R/kuiperllama_amd/lib/libkuiper_hip.so
  KH_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$v -o p -- python $R/tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 5 > $O/abl_$v.log 2>&1
  S=$(ls $O/abl_$v/*kernel_stats.csv $O/abl_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $v  $(grep prefill $O/abl_$v.log)"
  [ -n "$S" ] && grep "k_pg_\|k_attn" "$S" | awk -F, '{printf "   %-50s calls %s avg_ns %s min %s max %s\n", $1, $2, $4, $6, $7}'
  rm -rf $O/abl_$v
done
} > $O/r2_gemm_ablation_per_kernel.txt 2>&1
cat $O/r2_gemm_ablation_per_kernel.txt
