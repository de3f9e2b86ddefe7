#!/bin/bash
# Round 6: self-tests + mapped-on-demand KV cache: the new tests first, then the full GPU suite, the load probe with
# the cache mapped on demand vs plainly allocated (KH_KV_VMM=0), and the decode rate of both (same box).
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q --timeout=600 -k "selftests or mapped_on_demand" > $O/r6_new_tests.txt 2>&1
echo "new tests rc=$?" >> $O/r6_new_tests.txt
tail -25 $O/r6_new_tests.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/r6_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/r6_pytest_gpu.txt
grep -E "passed|failed|FAILED|rc=" $O/r6_pytest_gpu.txt | tail -15
P=$O/r6_load_probe.txt
: > $P
for w in llama3.2-1b llama2-7b-int8; do
  echo "== $w, KV cache mapped on demand" >> $P
  KH_LOAD_DEBUG=1 timeout 600 python tools/load_probe.py $w 2>&1 | grep -v amdgpu.ids >> $P
  echo "== $w, KH_KV_VMM=0 (one hipMalloc per cache, as rounds 1-5)" >> $P
  KH_KV_VMM=0 timeout 600 python tools/load_probe.py $w 2>&1 | grep -v "amdgpu.ids\|^\[kh" >> $P
done
grep -E "^==|\"load\"" $P | cut -c1-330
A=$O/r6_kv_vmm_ab.txt
: > $A
for i in 1 2; do
  for w in llama3.2-1b llama2-7b-int8; do
    python tools/kprof.py $w kv-mapped-on-demand 2>&1 | tail -1 | tee -a $A
    KH_KV_VMM=0 python tools/kprof.py $w kv-hipMalloc 2>&1 | tail -1 | tee -a $A
  done
done
