#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "embedding or prefill_api" > $O/r2_newtests.log 2>&1; echo "rc=$?" >> $O/r2_newtests.log
tail -5 $O/r2_newtests.log
bash tools/profile_round2.sh
