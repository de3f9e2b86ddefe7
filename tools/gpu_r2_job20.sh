#!/bin/bash
# decode prologue changes: parity tests of the decode path + bench
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "not full_size" > $O/r2_decode_tests.log 2>&1; echo "rc=$?" >> $O/r2_decode_tests.log
tail -4 $O/r2_decode_tests.log
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary']
print('fp32 %.1f tok/s' % d['value'], d['roofline']['kernels_avg_us'], '| int8 %.1f tok/s' % s['value'], s['roofline']['kernels_avg_us'])"; done
