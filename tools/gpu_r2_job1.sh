#!/bin/bash
# round-2 GPU job 1: box facts, chain microbenchmarks, full-size parity tests, int8 scale-prefetch A/B,
# merged-launch measurement on the small configs, counter list.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/memory.max; free -g | head -2; } > $O/r2_host.txt 2>&1
timeout 300 kuiperllama_amd/lib/mb_chain > $O/r2_mb_chain.txt 2>&1
(cd /tmp && TMPDIR=/tmp timeout 120 rocprofv3 -L > $O/r2_counters.txt 2>&1)
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -k "full_size" -x > $O/r2_fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/r2_fullsize.log
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_model_gpu.py::test_full_size_baseline_shapes > $O/r2_pytest_gpu.log 2>&1; echo "rc=$?" >> $O/r2_pytest_gpu.log
{
tools/run_env.sh llama2-7b-int8 "KH_Q8_SP=0" "KH_Q8_SP=1"
tools/run_env.sh qwen2.5-0.5b "KH_MERGE=0" "KH_MERGE=1"
tools/run_env.sh tinyllama-1.1b "KH_MERGE=0" "KH_MERGE=1"
} > $O/r2_env_ab.txt 2>&1
tail -5 $O/r2_fullsize.log; tail -3 $O/r2_pytest_gpu.log; cat $O/r2_env_ab.txt; head -60 $O/r2_mb_chain.txt
