#!/bin/bash
# Round 6, VERDICT r5 item 8: the Llama-3.2-1B ffn13 decode GEMV on the matrix cores vs the shipped VALU kernel
# (tools/mb_gemv_mfma.hip), then the HBM read bytes (FETCH_SIZE) and the MFMA / VALU utilisation of each form in
# separate rocprofv3 --pmc passes (--kernel-trace only beside them).
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
OUT=$O/r6_gemv_mfma_ab.txt
B=$PWD/kuiperllama_amd/lib/mb_gemv_mfma
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mb_gemv_mfma.hip -o $B
: > $OUT
for i in 1 2 3; do timeout 120 $B 16 20 2>&1 | grep -v amdgpu.ids | tee -a $OUT; done
export TMPDIR=/tmp; cd /tmp
for c in "FETCH_SIZE" "MfmaUtil" "VALUBusy" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  d=$O/pmc_gemv_$(echo $c | tr ' ' '_' | cut -c1-24)
  rm -rf $d
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $d -o m -- $B 16 2 > $O/pmc_gemv_log.txt 2>&1 || echo "pass $c exit $?"
done
cd $O/..
python - >> $OUT <<'PY'
import glob, re, sqlite3
print("\n## rocprofv3 --pmc passes (per launch, averages over the launches of `mb_gemv_mfma 16 2`)")
rows = {}
for db in sorted(glob.glob("gpurun_out/pmc_gemv_*/*results.db")):
    try:
        con = sqlite3.connect(db)
        q = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name")
    except Exception as e:
        print("skip", db, e); continue
    for name, cn, n, avg in q:
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))
        if k.startswith("k_ffn13"):
            rows.setdefault(k, {})[cn] = avg
alg = 2 * 8192 * 2048 * 4 + 2 * 2048 * 4 + 8192 * 4
for k, v in sorted(rows.items()):
    extra = ""
    if "FETCH_SIZE" in v:
        b = v["FETCH_SIZE"] * 1024 * 2  # KiB, gfx950 half-count of a 16 B/lane stream (MI355X_MICROARCH.md, HBM section)
        extra = f"  HBM read {b / 1e6:.1f} MB = {b / alg:.3f} x algorithmic"
    print(f"{k:28s} " + "  ".join(f"{c} {x:.4g}" for c, x in sorted(v.items())) + extra)
PY
rm -rf $O/pmc_gemv_*/
tail -25 $OUT
