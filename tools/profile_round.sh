#!/bin/bash
# The profile set of a round (run on the GPU box through gpurun; ROUND=r6 KH_COMMIT=<hash> in the environment);
# summaries land in gpurun_out/ AND in profiles/ of the box's copy - copy gpurun_out/<round>_* into profiles/:
#   1. rocprofv3 --kernel-trace --stats of the bench command (--no-others: the other configs launch
#      the same kernel instantiations at other sizes and would blur the averages; no CPU baseline)
#   2. FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel-trace only) on the lean workload
#   3. utilisation counters (VALUBusy, MfmaUtil, occupancy, SQ wait/active cycles) for the fp32 and
#      the int8 workload, decode steps + one GEMM prefill
# (rounds 4 and 5: `git show 87acefa:tools/profile_round5.sh`)
ROUND=${ROUND:-r6}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/pf_stats -o bench -- python $R/bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-others > $O/${ROUND}_bench_under_rocprof.json 2> $O/${ROUND}_pf_stats.log || echo "stats pass exit $?"
for w in llama3.2-1b llama2-7b-int8; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pf_${c}_$w -o bench -- python $R/tools/pmc_workload.py $w --steps 8 > $O/pf_${c}_$w.log 2>&1 || echo "$c $w exit $?"
done; done
cd $R
DB=$(ls gpurun_out/pf_stats/*results.db gpurun_out/pf_stats/*/*results.db 2>/dev/null | head -1)
python tools/rocpd_timeline.py $DB > gpurun_out/${ROUND}_kernel_timeline.txt 2>&1
CMD="tools/profile_round.sh (ROUND=$ROUND): rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python tools/pmc_workload.py <workload> --steps 8"
for w in llama3.2-1b llama2-7b-int8; do
  F=$(ls gpurun_out/pf_FETCH_SIZE_$w/*results.db gpurun_out/pf_FETCH_SIZE_$w/*/*results.db 2>/dev/null | head -1)
  W=$(ls gpurun_out/pf_WRITE_SIZE_$w/*results.db gpurun_out/pf_WRITE_SIZE_$w/*/*results.db 2>/dev/null | head -1)
  if [ "$w" = llama3.2-1b ]; then python tools/rocpd_summary.py --round $ROUND --stats $DB --fetch $F --write $W --workload $w --command "$CMD"
  else python tools/rocpd_summary.py --round $ROUND --fetch $F --write $W --workload $w --command "$CMD"; fi
  cp profiles/${ROUND}_pmc.csv gpurun_out/${ROUND}_pmc_$w.csv
done
rm -f profiles/${ROUND}_pmc.csv
cp profiles/${ROUND}_kernel_stats.csv profiles/pmc_traffic.json gpurun_out/
rm -rf gpurun_out/pf_stats gpurun_out/pf_FETCH* gpurun_out/pf_WRITE*
tools/profile_pmc.sh llama3.2-1b $O/${ROUND}_pmc_util_1b.csv --steps 8 --prefill gemm > $O/${ROUND}_pmc_1b.log 2>&1
tools/profile_pmc.sh llama2-7b-int8 $O/${ROUND}_pmc_util_int8.csv --steps 8 --prefill gemm > $O/${ROUND}_pmc_int8.log 2>&1
head -16 gpurun_out/${ROUND}_kernel_stats.csv; head -c 600 $O/${ROUND}_bench_under_rocprof.json; echo; head -12 gpurun_out/${ROUND}_kernel_timeline.txt
