#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "gemm or prefill_api" -x > $O/r2_gemm.log 2>&1; echo "gemm rc=$?" >> $O/r2_gemm.log
for w in llama3.2-1b llama2-7b-int8 qwen2.5-0.5b tinyllama-1.1b; do echo "== $w"; timeout 300 python tools/pmc_workload.py $w --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill; done > $O/r2_prefill_speed.txt 2>&1
{ KH_PG_DEBUG=1 timeout 200 python tools/pmc_workload.py llama2-7b-int8 --steps 0 --prefill gemm --reps 1 2>&1 | grep "\[pg\]" | sort | uniq -c; } > $O/r2_pg_shapes_7b.txt 2>&1
tail -4 $O/r2_gemm.log; cat $O/r2_prefill_speed.txt $O/r2_pg_shapes_7b.txt
