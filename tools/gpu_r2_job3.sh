#!/bin/bash
# round-2 GPU job 3: GEMM prefill parity after the phased K loop, kernel-trace stats of the prefill,
# counter passes on the lean workload.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "gemm" -x -s > $O/r2_gemm.log 2>&1; echo "gemm rc=$?" >> $O/r2_gemm.log
for w in llama3.2-1b llama2-7b-int8; do timeout 300 python tools/pmc_workload.py $w --steps 0 --prefill gemm,gemv --reps 3; done > $O/r2_prefill_speed.txt 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/pf_stats -o pf -- python $R/tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 > $O/r2_pf_stats.log 2>&1
cd $R
python - <<'PY' > gpurun_out/r2_prefill_kernel_stats.txt 2>&1
import sqlite3, glob
for db in glob.glob("gpurun_out/pf_stats/**/*results.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    k = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
    print(db, k[:5])
    try:
        for r in con.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, max(grid_x), max(workgroup_x), max(vgpr_count), max(lds_size) from kernels group by name, grid_x order by 3 desc limit 12"):
            print("%-60s n=%5d total_us=%10.1f avg_us=%8.2f grid=%d wg=%d vgpr=%d lds=%d" % (r[0][:60], *r[1:]))
    except Exception as e:
        print("ERR", e, tabs[:20])
PY
rm -rf gpurun_out/pf_stats
timeout 600 tools/profile_pmc.sh llama3.2-1b $O/r2_pmc_util_1b.csv --steps 8 --prefill gemm > $O/r2_pmc_1b.log 2>&1
tail -25 $O/r2_gemm.log; cat $O/r2_prefill_speed.txt; cat $O/r2_prefill_kernel_stats.txt; grep "k_pg_gemm\|^kernel" $O/r2_pmc_util_1b.csv
