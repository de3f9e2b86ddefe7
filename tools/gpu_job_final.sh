#!/bin/bash
# final check of a round: full GPU suite, smoke(), the default bench command (as the driver runs it)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/r3_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r3_smoke.txt 2>&1
echo "smoke rc=$?" >> $O/r3_smoke.txt
timeout 900 python bench.py > $O/r3_bench.json 2> $O/r3_bench.err
echo "bench rc=$?" >> $O/r3_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r3_bench_steps20.json 2> $O/r3_bench_steps20.err
tail -4 $O/r3_pytest_gpu.txt; tail -2 $O/r3_smoke.txt; grep -E "bench rc" $O/r3_bench.err; head -c 400 $O/r3_bench.json
