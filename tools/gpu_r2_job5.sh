#!/bin/bash
# round-2 GPU job 5: balanced GEMM shapes: parity, speed, then the whole GPU suite and the bench line
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "gemm" -x -s > $O/r2_gemm.log 2>&1; echo "gemm rc=$?" >> $O/r2_gemm.log
for w in llama3.2-1b llama2-7b-int8 qwen2.5-0.5b tinyllama-1.1b; do echo "== $w"; KH_PG_DEBUG=${PGDBG:-} timeout 300 python tools/pmc_workload.py $w --steps 0 --prefill gemm,gemv --reps 3 2>&1 | grep prefill; done > $O/r2_prefill_speed.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r2_pytest_gpu.log 2>&1; echo "rc=$?" >> $O/r2_pytest_gpu.log
timeout 600 python bench.py > $O/r2_bench.json 2> $O/r2_bench.err; echo "bench rc=$?" >> $O/r2_bench.err
tail -12 $O/r2_gemm.log; cat $O/r2_prefill_speed.txt; tail -4 $O/r2_pytest_gpu.log; tail -2 $O/r2_bench.err
