#!/bin/bash
# after the workgroup-size fix: full GPU suite + bench
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $O/r2_pytest_gpu.log 2>&1; echo "rc=$?" >> $O/r2_pytest_gpu.log
tail -6 $O/r2_pytest_gpu.log
timeout 900 python bench.py > $O/r2_bench.json 2> $O/r2_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r2_bench.json").read().strip().splitlines()[-1]); s=d["secondary"]
print("fp32 %.1f tok/s" % d["value"], d["roofline"]["frac"], d["roofline"]["step"]["frac"], d["roofline"]["kernels_avg_us"])
print("int8 %.1f tok/s" % s["value"], s["roofline"]["step"]["frac"], s["roofline"]["kernels_avg_us"])
print(d["prefill"]); print(s.get("prefill")); print(d["cpu_baseline"]["tokens_match_gpu"], d["cpu_baseline"]["tokens_compared"])
P
