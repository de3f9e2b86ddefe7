#!/bin/bash
# round 4, call A: the GPU suite on the deferred split merge, the same-box A/B of both merge forms at deep
# positions (three models), the switch point of the GQA group path re-swept, the default bench line.
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $O/r4_pytest_gpu_a.txt 2>&1
echo "pytest rc=$?" >> $O/r4_pytest_gpu_a.txt
tail -5 $O/r4_pytest_gpu_a.txt
{
  timeout 600 python tools/attn_defer_ab.py llama3.2-1b qwen2.5-0.5b llama2-7b-int8
  for tl in 8192 16384 32768; do KH_ATTN_TLONG=$tl timeout 300 python tools/attn_defer_ab.py llama3.2-1b; done
} 2>&1 | grep -v amdgpu.ids > $O/r4_attn_defer_ab.txt
tail -30 $O/r4_attn_defer_ab.txt
timeout 900 python bench.py > $O/r4_bench_a.json 2> $O/r4_bench_a.err
echo "bench rc=$?" >> $O/r4_bench_a.err
tail -3 $O/r4_bench_a.err; head -c 600 $O/r4_bench_a.json
