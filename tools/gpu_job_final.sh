#!/bin/bash
# final check of a round: full GPU suite, smoke(), the default bench command (as the driver runs it) and the
# driver's short form; outputs under gpurun_out/<ROUND>_* (ROUND=r6 by default)
cd "$(dirname "$0")/.."
ROUND=${ROUND:-r6}
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/${ROUND}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/${ROUND}_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${ROUND}_smoke.txt 2>&1
echo "smoke rc=$?" >> $O/${ROUND}_smoke.txt
SECONDS=0
timeout 900 python bench.py > $O/${ROUND}_bench.json 2> $O/${ROUND}_bench.err
echo "bench rc=$? wall=${SECONDS}s" >> $O/${ROUND}_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/${ROUND}_bench_steps20.json 2> $O/${ROUND}_bench_steps20.err
timeout 300 python tools/gen_overhead.py 2>&1 | grep -v amdgpu > $O/${ROUND}_generate_overhead.txt
tail -4 $O/${ROUND}_pytest_gpu.txt; tail -2 $O/${ROUND}_smoke.txt; grep -E "bench rc" $O/${ROUND}_bench.err; head -c 400 $O/${ROUND}_bench.json
