#!/bin/bash
# insurance: the GPU suite three times in a row under the native-backtrace preload (tools/dbg/segv_bt.c)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
[ -f tools/dbg/libsegv_bt.so ] || gcc -shared -fPIC -O1 -o tools/dbg/libsegv_bt.so tools/dbg/segv_bt.c
export LD_PRELOAD=$PWD/tools/dbg/libsegv_bt.so
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:faulthandler --timeout=900 > $O/r6_suite_rep_$i.txt 2>&1
  echo "pytest rc=$?" >> $O/r6_suite_rep_$i.txt
  grep -E "passed|failed|rc=" $O/r6_suite_rep_$i.txt | tail -2
done
