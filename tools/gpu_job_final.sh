#!/bin/bash
# final check of a round: full GPU suite, smoke(), the default bench command (as the driver runs it) and the
# driver's short form; outputs under gpurun_out/r5_*
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/r5_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/r5_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r5_smoke.txt 2>&1
echo "smoke rc=$?" >> $O/r5_smoke.txt
SECONDS=0
timeout 900 python bench.py > $O/r5_bench.json 2> $O/r5_bench.err
echo "bench rc=$? wall=${SECONDS}s" >> $O/r5_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r5_bench_steps20.json 2> $O/r5_bench_steps20.err
timeout 300 python tools/gen_overhead.py 2>&1 | grep -v amdgpu > $O/r5_generate_overhead.txt
tail -4 $O/r5_pytest_gpu.txt; tail -2 $O/r5_smoke.txt; grep -E "bench rc" $O/r5_bench.err; head -c 400 $O/r5_bench.json
