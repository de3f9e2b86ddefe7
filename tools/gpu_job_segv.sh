#!/bin/bash
# stress the create / destroy cycle with a native backtrace on a crash (tools/dbg/segv_bt.c):
# address ranges kept (default), freed with hipMemAddressFree (KH_KV_VA_POOL=0: crashes), plain allocation
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
[ -f tools/dbg/libsegv_bt.so ] || gcc -shared -fPIC -O1 -o tools/dbg/libsegv_bt.so tools/dbg/segv_bt.c
export LD_PRELOAD=$PWD/tools/dbg/libsegv_bt.so
F=$O/r6_vmm_destroy_crash.txt; : > $F
run() { echo "== $*" >> $F; ( env "$@" timeout 500 python tools/stress_destroy.py $T 2>&1; echo "rc=$?" ) | grep -v -E "^Extension|amdgpu.ids|^python\(|libffi|_ctypes" | cut -c1-200 >> $F; }
T=300 run KH_NOTHING=1
T=150 run KH_KV_VA_POOL=0
T=60 run KH_KV_VMM=0
cat $F
