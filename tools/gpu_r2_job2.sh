#!/bin/bash
# round-2 GPU job 2: GEMM prefill parity + adapter binding tests, whole suite, bench line, int8 PMC.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "gemm" -x > $O/r2_gemm.log 2>&1; echo "gemm rc=$?" >> $O/r2_gemm.log
timeout 600 python -m pytest tests/test_cpp_adapter.py tests/test_abi.py -q -m gpu > $O/r2_adapter.log 2>&1; echo "adapter rc=$?" >> $O/r2_adapter.log
timeout 600 python bench.py > $O/r2_bench.json 2> $O/r2_bench.err; echo "bench rc=$?" >> $O/r2_bench.err
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_model_gpu.py::test_full_size_baseline_shapes > $O/r2_pytest_gpu.log 2>&1; echo "rc=$?" >> $O/r2_pytest_gpu.log
timeout 1200 tools/profile_pmc.sh llama2-7b-int8 $O/r2_pmc_util_int8.csv > $O/r2_pmc_int8.log 2>&1
tail -30 $O/r2_gemm.log; tail -5 $O/r2_adapter.log; tail -3 $O/r2_pytest_gpu.log; tail -3 $O/r2_bench.err; head -c 3000 $O/r2_bench.json; tail -40 $O/r2_pmc_int8.log
