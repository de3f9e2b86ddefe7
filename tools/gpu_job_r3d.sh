#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O; R=$PWD
{
  for i in 1 2; do
    KH_LIB=$R/kuiperllama_amd/lib/exp_r2.so timeout 300 python tools/prefill_time.py r2 llama3.2-1b tinyllama-1.1b qwen2.5-0.5b
    timeout 300 python tools/prefill_time.py r3 llama3.2-1b tinyllama-1.1b qwen2.5-0.5b
  done
  KH_PG_DEBUG=1 timeout 300 python tools/prefill_time.py r3dbg llama3.2-1b 2>&1 | sort | uniq -c | sort -rn | head -12
} > $O/r3_prefill_ab.txt 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_prefill -o pf -- python $R/tools/prefill_time.py r3 llama3.2-1b > $R/$O/r3_prefill_rocprof.log 2>&1)
python - <<'PY' > $O/r3_prefill_kernel_stats.txt 2>&1
import csv, glob
fs = glob.glob('gpurun_out/prof_prefill/**/*kernel_stats.csv', recursive=True)
for f in fs:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get('TotalDurationNs', r.get('Total Duration (ns)', 0)) or 0))
    for r in rows[:14]:
        print({k: r[k] for k in list(r)[:6]})
PY
timeout 900 python -m pytest tests -m gpu -q -k "prefill" --timeout=900 > $O/r3_pytest_prefill.txt 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_prefill.txt
grep -v amdgpu.ids $O/r3_prefill_ab.txt; cat $O/r3_prefill_kernel_stats.txt; tail -5 $O/r3_pytest_prefill.txt
