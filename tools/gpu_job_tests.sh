#!/bin/bash
# full GPU suite + smoke (no -x: every failure of the round in one call)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/r4_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/r4_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r4_smoke.txt 2>&1
echo "smoke rc=$?" >> $O/r4_smoke.txt
grep -E "passed|failed|FAILED|rc=" $O/r4_pytest_gpu.txt | tail -15; tail -2 $O/r4_smoke.txt
