#!/bin/bash
# usage: tools/profile_pmc.sh <workload> <out.csv> [pmc_workload.py args]
# Per-kernel utilisation counters of the decode step (and of the GEMM prefill bench.py also runs),
# separate --pmc passes with --kernel-trace only (gpurun refuses --pmc beside the trace domains).
# Derived: VALUBusy, MfmaUtil, OccupancyPercent, MemUnitStalled ; the gfx94x "MemUnitBusy" formula
# evaluates to nan on gfx950, the texture-addresser busy time (TA_BUSY_avr / GRBM_GUI_ACTIVE) stands in.
W=$1; OUT=$2; shift 2
R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
PASSES=("VALUBusy" "MfmaUtil" "OccupancyPercent" "MemUnitStalled" "TA_BUSY_avr GRBM_GUI_ACTIVE" \
        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
        "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT")
i=0
for c in "${PASSES[@]}"; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$i -o b -- python $R/tools/pmc_workload.py $W "$@" > $R/gpurun_out/pmclog_$i.txt 2>&1 || echo "pass $i ($c) exit $?"
  i=$((i+1))
done
cd $R
python - "$OUT" <<'PY'
import sqlite3, glob, re, csv, sys
rows, cols = {}, []
for db in sorted(glob.glob("gpurun_out/pmc_*/*results.db")):
    try:
        con = sqlite3.connect(db)
        q = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                        "group by kernel_name, counter_name")
    except Exception as e:
        print("skip", db, e); continue
    for name, cn, n, avg in q:
        if not re.match(r"^(void )?k_[a-z0-9_]+", name):
            continue
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))
        rows.setdefault(k, {})[cn] = (n, avg)
        if cn not in cols: cols.append(cn)
with open(sys.argv[1], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches"] + cols)
    for k, v in sorted(rows.items()):
        w.writerow([k, max(x[0] for x in v.values())] + [f"{v[c][1]:.4g}" if c in v else "" for c in cols])
print(open(sys.argv[1]).read())
PY
rm -rf gpurun_out/pmc_[0-9]*
