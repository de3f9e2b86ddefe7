#!/bin/bash
# Round 6: the full GPU suite (no -x: every failure in one call) + smoke, then a targeted launch-shape sweep of the
# Llama-3.2-1B decode GEMVs under the exact staging depth (tools/sweep_quick.py; the round-4 sweep ran with two
# dummy slots per thread in the 256-thread shapes and three in the 512-thread ones, profiles/r4_shape_sweep_1b.txt).
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/r6_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/r6_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.txt 2>&1
echo "smoke rc=$?" >> $O/r6_smoke.txt
grep -E "passed|failed|FAILED|rc=" $O/r6_pytest_gpu.txt | tail -15; tail -2 $O/r6_smoke.txt
timeout 900 python tools/sweep_quick.py llama3.2-1b \
  "QKV=2,4,384,512;1,8,384,256;1,8,192,512;2,4,512,256;1,4,384,256" \
  "WO=2,4,256,512;1,8,256,256;4,2,512,512;4,2,1024,256;1,8,128,512" \
  "FFN=1,8,256,512;1,8,1024,256;1,8,512,512;1,4,512,256" \
  "CLS=1,8,512,256;1,8,512,512;1,8,1024,256" 2>&1 | grep -v amdgpu.ids > $O/r6_shape_sweep_1b.txt
tail -30 $O/r6_shape_sweep_1b.txt
