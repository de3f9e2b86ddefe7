#!/bin/bash
# Final round-2 evidence set: full GPU suite, smoke, the driver's bench commands, the profile set.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $O/r2_pytest_gpu.log 2>&1; echo "rc=$?" >> $O/r2_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_smoke.log 2>&1; echo "rc=$?" >> $O/r2_smoke.log
timeout 900 python bench.py > $O/r2_bench.json 2> $O/r2_bench.err; echo "bench rc=$?" >> $O/r2_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r2_bench_steps20.json 2> $O/r2_bench_steps20.err; echo "bench20 rc=$?" >> $O/r2_bench_steps20.err
bash tools/profile_round2.sh > $O/r2_profile.log 2>&1
export TMPDIR=/tmp; cd /tmp
for w in llama3.2-1b llama2-7b-int8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pfk_$w -o p -- python $R/tools/pmc_workload.py $w --steps 0 --prefill gemm --reps 5 > $O/pfk_$w.log 2>&1
  S=$(ls $O/pfk_$w/*kernel_stats.csv $O/pfk_$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$S" ] && head -14 "$S" > $O/r2_prefill_kernel_stats_$w.csv
  rm -rf $O/pfk_$w
done
cd $R
tail -3 $O/r2_pytest_gpu.log; cat $O/r2_smoke.log | tail -2; tail -2 $O/r2_bench.err; head -c 600 $O/r2_bench_steps20.json; echo; cat $O/r2_prefill_kernel_stats_llama3.2-1b.csv
