#!/bin/bash
# stress the create / destroy cycle with a native backtrace on a crash (tools/dbg/segv_bt.c)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
export LD_PRELOAD=$PWD/tools/dbg/libsegv_bt.so
timeout 400 python tools/stress_destroy.py 240 > $O/r6_dbg_stress_vmm.txt 2>&1; echo "rc=$?" >> $O/r6_dbg_stress_vmm.txt
KH_KV_VMM=0 timeout 300 python tools/stress_destroy.py 120 > $O/r6_dbg_stress_plain.txt 2>&1; echo "rc=$?" >> $O/r6_dbg_stress_plain.txt
grep -v "^Extension" $O/r6_dbg_stress_vmm.txt | tail -30; tail -5 $O/r6_dbg_stress_plain.txt
