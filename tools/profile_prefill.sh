R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_prof -o pf -- python $R/tools/bench_prefill.py ${1:-llama3.2-1b} > $R/gpurun_out/pf_prof.log 2>&1
cd $R
python - <<'PY'
import sqlite3,glob,re
db=sqlite3.connect(glob.glob('gpurun_out/pf_prof/*results.db')[0])
rows=db.execute("select name,count(*),avg(duration),sum(duration) from kernels group by name order by sum(duration) desc").fetchall()
for n,c,a,t in rows[:14]:
    n=re.sub(r'\(.*$','',re.sub(r'^void ','',n))
    print(f"{n:50s} {c:7d} {a/1e3:9.2f} us {t/1e6:9.2f} ms")
PY
rm -rf gpurun_out/pf_prof
