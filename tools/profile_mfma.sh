#!/bin/bash
# MFMA / VALU utilisation of the decode kernels (north star asks for MFMA-busy; the design keeps
# the GEMVs on the VALU).  Separate --pmc passes, kernel-trace only.
R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for c in MfmaUtil VALUBusy MemUnitBusy; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pm_$c -o b -- python $R/bench.py --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --secondary "" > /dev/null 2> $R/gpurun_out/pm_$c.log
done
cd $R
python - <<'PY'
import sqlite3, glob, re, csv
rows = {}
for c in ("MfmaUtil", "VALUBusy", "MemUnitBusy"):
    for db in glob.glob(f"gpurun_out/pm_{c}/*results.db"):
        con = sqlite3.connect(db)
        for name, cn, n, avg in con.execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                "group by kernel_name, counter_name"):
            if not re.match(r"^(void )?k_[a-z0-9_]+", name):
                continue
            k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))
            rows.setdefault(k, {})[cn] = (n, avg)
with open("gpurun_out/r1_pmc_util.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "MfmaUtil_pct", "VALUBusy_pct", "MemUnitBusy_pct", "dispatches"])
    for k, v in sorted(rows.items()):
        w.writerow([k] + [f"{v.get(c, (0, float('nan')))[1]:.2f}" for c in ("MfmaUtil", "VALUBusy", "MemUnitBusy")]
                   + [max(x[0] for x in v.values())])
print(open("gpurun_out/r1_pmc_util.csv").read())
PY
rm -rf gpurun_out/pm_*
