R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_stats -o bench -- python $R/bench.py --steps 128 --warmup 16 --no-cpu-baseline > $R/gpurun_out/pf_bench.json 2> $R/gpurun_out/pf_stats.log
for w in llama3.2-1b llama2-7b-int8; do for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pf_${c}_$w -o bench -- python $R/bench.py --workload $w --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --secondary "" > /dev/null 2> $R/gpurun_out/pf_${c}_$w.log
done; done
cd $R
python tools/rocpd_timeline.py gpurun_out/pf_stats/bench_results.db > gpurun_out/timeline.txt
python tools/rocpd_summary.py --round r1 --stats gpurun_out/pf_stats/bench_results.db --fetch gpurun_out/pf_FETCH_SIZE_llama3.2-1b/bench_results.db --write gpurun_out/pf_WRITE_SIZE_llama3.2-1b/bench_results.db --workload llama3.2-1b
cp profiles/r1_pmc.csv gpurun_out/r1_pmc_llama3.2-1b.csv
python tools/rocpd_summary.py --round r1 --fetch gpurun_out/pf_FETCH_SIZE_llama2-7b-int8/bench_results.db --write gpurun_out/pf_WRITE_SIZE_llama2-7b-int8/bench_results.db --workload llama2-7b-int8
cp profiles/r1_pmc.csv gpurun_out/r1_pmc_llama2-7b-int8.csv
cp profiles/r1_kernel_stats.csv profiles/pmc_traffic.json gpurun_out/
rm -rf gpurun_out/pf_stats gpurun_out/pf_FETCH* gpurun_out/pf_WRITE*
head -12 gpurun_out/timeline.txt; cat gpurun_out/pf_bench.json | head -c 600
