#!/bin/bash
# 1024-thread workgroups for the GEMV decode kernels (x staged once per CU): variant build + shape overrides; RCCL test
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_replicas_gloo.py -q -m gpu -x 2>&1 | tail -3
X=$R/kuiperllama_amd/lib/exp_wg1024.so
{
echo "## llama2-7b-int8"
tools/run_env.sh llama2-7b-int8 "KH_X=0" "KH_LIB=$X" \
  "KH_LIB=$X KH_SHAPE_W2=4,4,256,1024" "KH_LIB=$X KH_SHAPE_W2=2,4,256,1024" "KH_LIB=$X KH_SHAPE_W2=4,2,256,1024" "KH_LIB=$X KH_SHAPE_W2=4,4,512,1024" \
  "KH_LIB=$X KH_SHAPE_QKV=1,4,256,1024" "KH_LIB=$X KH_SHAPE_QKV=1,2,256,1024" "KH_LIB=$X KH_SHAPE_QKV=2,2,256,1024" \
  "KH_LIB=$X KH_SHAPE_FFN=1,4,256,1024" "KH_LIB=$X KH_SHAPE_WO=2,2,256,1024" "KH_LIB=$X KH_SHAPE_WO=4,2,256,1024" "KH_LIB=$X KH_SHAPE_CLS=1,4,256,1024"
echo "## llama3.2-1b"
tools/run_env.sh llama3.2-1b "KH_X=0" "KH_LIB=$X" \
  "KH_LIB=$X KH_SHAPE_W2=4,8,256,1024" "KH_LIB=$X KH_SHAPE_W2=4,4,256,1024" "KH_LIB=$X KH_SHAPE_W2=2,8,256,1024" \
  "KH_LIB=$X KH_SHAPE_QKV=2,4,256,1024" "KH_LIB=$X KH_SHAPE_QKV=1,4,256,1024" "KH_LIB=$X KH_SHAPE_FFN=1,8,256,1024" "KH_LIB=$X KH_SHAPE_FFN=1,4,256,1024" \
  "KH_LIB=$X KH_SHAPE_WO=2,4,256,1024" "KH_LIB=$X KH_SHAPE_WO=4,4,256,1024" "KH_LIB=$X KH_SHAPE_CLS=1,8,256,1024"
} > $O/r2_wg1024.txt 2>&1
cat $O/r2_wg1024.txt
