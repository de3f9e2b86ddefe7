#!/bin/bash
R=$PWD; O=$R/gpurun_out
for e in KH_ATTN_WG=512 KH_ATTN_WG=256; do echo -n "[$e] "; env $e timeout 120 python tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/pf_stats -o pf -- python $R/tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 > $O/r2_pf_stats.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for db in glob.glob("gpurun_out/pf_stats/**/*results.db", recursive=True):
    con = sqlite3.connect(db)
    for r in con.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, max(grid_x), max(workgroup_x) from kernels group by name, grid_x order by 3 desc limit 9"):
        print("%-58s n=%5d total_us=%9.1f avg_us=%8.2f grid=%d wg=%d" % (r[0][:58], *r[1:]))
PY
rm -rf gpurun_out/pf_stats
