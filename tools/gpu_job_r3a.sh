#!/bin/bash
# round 3, GPU call A: new parity tests, first A/B of the rolling-refill kernels vs the r2 library,
# int8 stream floors, the default bench line
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
R=$PWD
{
  echo "== int8 floors"; timeout 300 ./kuiperllama_amd/lib/mb_int8_floors
} > $O/r3_int8_floors.txt 2>&1
{
  for w in llama2-7b-int8 llama3.2-1b; do
    for i in 1 2; do
      KH_LIB=$R/kuiperllama_amd/lib/exp_r2.so timeout 300 python tools/kprof.py $w r2
      timeout 300 python tools/kprof.py $w r3
    done
  done
  for w in qwen2.5-0.5b tinyllama-1.1b; do
    KH_LIB=$R/kuiperllama_amd/lib/exp_r2.so timeout 300 python tools/kprof.py $w r2
    timeout 300 python tools/kprof.py $w r3
  done
} > $O/r3_ab_rolling.txt 2>$O/r3_ab_rolling.err
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $O/r3_pytest_gpu_a.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_gpu_a.txt
timeout 600 python bench.py > $O/r3_bench_a.json 2> $O/r3_bench_a.err
echo "bench rc=$?" >> $O/r3_bench_a.err
tail -5 $O/r3_int8_floors.txt; cat $O/r3_ab_rolling.txt; tail -15 $O/r3_pytest_gpu_a.txt; tail -3 $O/r3_bench_a.err
