#!/bin/bash
# refresh the profile set with the final build (bench under rocprof, PMC traffic + utilisation, prefill kernel stats)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/profile_round2.sh > $O/r2_profile.log 2>&1
export TMPDIR=/tmp; cd /tmp
for w in llama3.2-1b llama2-7b-int8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pfk_$w -o p -- python $R/tools/pmc_workload.py $w --steps 0 --prefill gemm --reps 5 > $O/pfk_$w.log 2>&1
  S=$(ls $O/pfk_$w/*kernel_stats.csv $O/pfk_$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$S" ] && grep "Name\|k_pg_\|k_attn\|k_embedding" "$S" > $O/r2_prefill_kernel_stats_$w.csv
  rm -rf $O/pfk_$w
done
cd $R
{ for w in llama3.2-1b llama2-7b-int8 qwen2.5-0.5b tinyllama-1.1b; do echo "== $w"; timeout 300 python tools/pmc_workload.py $w --steps 0 --prefill gemm --reps 3 2>&1 | grep prefill; done
  for pl in "1024 0" "128 1920" "128 8064"; do set -- $pl; echo -n "== llama3.2-1b prompt $1 pos0 $2  "; timeout 300 python tools/pmc_workload.py llama3.2-1b --steps 0 --prefill gemm --reps 3 --prompt $1 --pos0 $2 2>&1 | grep prefill; done; } > $O/r2_prefill_speed.txt 2>&1
for w in llama2-7b qwen2.5-0.5b tinyllama-1.1b; do timeout 900 python bench.py --workload $w --secondary "" --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json; done
tail -5 $O/r2_profile.log; cat $O/r2_prefill_speed.txt; grep "pg_\|kernel" $O/r2_pmc_util_1b.csv | cut -d, -f1-5
for w in llama2-7b qwen2.5-0.5b tinyllama-1.1b; do python -c "
import json;d=json.loads(open('$O/bench_$w.json').read());print('$w',round(d['value'],1),round(d['ms_per_step'],3),round(d['roofline']['step']['frac'],3),d['roofline']['frac'])"; done
