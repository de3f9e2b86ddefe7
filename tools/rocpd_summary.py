#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the small, tracked files under profiles/.

  python tools/rocpd_summary.py --round r1 --stats gpurun_out/prof_stats/bench_results.db \
         [--fetch gpurun_out/prof_fetch/bench_results.db] [--write gpurun_out/prof_write/...db] \
         [--workload llama3.2-1b]

Writes profiles/<round>_kernel_stats.csv (the `rocprofv3 --kernel-trace --stats` view restricted
to this library's kernels), profiles/<round>_pmc.csv and profiles/pmc_traffic.json.

PMC units / gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced stream,
so read bytes = FETCH_SIZE * 1024 * 2.  WRITE_SIZE is uncalibrated and reported raw (KiB*1024).
"""
import argparse
import csv
import json
import os
import sys
import re
import sqlite3

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def ours(name: str) -> bool:
    return bool(re.match(r"^(void )?k_[a-z0-9_]+", name))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r1")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--workload", default="llama3.2-1b")
    ap.add_argument("--commit", default=os.environ.get("KH_COMMIT", ""),
                    help="commit the counters were collected on (stamped into pmc_traffic.json)")
    ap.add_argument("--command", default="tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE "
                                         "--kernel-trace -- python tools/pmc_workload.py <workload> --steps 8")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)

    if a.stats:
        db = sqlite3.connect(a.stats)
        rows = db.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
            "max(grid_x), max(workgroup_x), max(vgpr_count), max(lds_size) "
            "from kernels group by name, grid_x, workgroup_x").fetchall()
        mine = [r for r in rows if ours(r[0])]
        tot = sum(r[2] for r in mine)
        p = os.path.join(out, f"{a.round}_kernel_stats.csv")
        with open(p, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_ours",
                        "grid_x_threads", "wg", "vgpr", "lds_bytes"])
            for r in sorted(mine, key=lambda r: -r[2]):
                w.writerow([short(r[0]), r[1], f"{r[2] / 1e3:.1f}", f"{r[3] / 1e3:.3f}",
                            f"{r[4] / 1e3:.3f}", f"{r[5] / 1e3:.3f}", f"{100 * r[2] / tot:.2f}",
                            r[6], r[7], r[8], r[9]])
        print("wrote", p)

    traffic = {}
    pmc_rows = []
    for path, counter in ((a.fetch, "FETCH_SIZE"), (a.write, "WRITE_SIZE")):
        if not path:
            continue
        db = sqlite3.connect(path)
        for name, cnt, n, avg, mn, mx in db.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                "from counters_collection group by kernel_name, counter_name"):
            if not ours(name) or cnt != counter:
                continue
            pmc_rows.append([short(name), cnt, n, f"{avg:.3f}", f"{mn:.3f}", f"{mx:.3f}"])
            # k_ffn13_ring / k_cls_ring are the LDS-DMA ring forms of the same launches: same key as the kernel they replace
            key = short(name).split("<")[0].replace("k_", "").replace("_ring", "")
            t = traffic.setdefault(f"{a.workload}:{key}", {})
            if counter == "FETCH_SIZE":
                t["read_bytes"] = avg * 1024 * 2  # gfx950: FETCH_SIZE counts 128-B requests as 64 B
                t["fetch_size_kib_raw"] = avg
            else:
                t["write_bytes_uncalibrated"] = avg * 1024
    if pmc_rows:
        p = os.path.join(out, f"{a.round}_pmc.csv")
        with open(p, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "avg_kib", "min_kib", "max_kib"])
            w.writerows(sorted(pmc_rows))
        print("wrote", p)
        tp = os.path.join(out, "pmc_traffic.json")
        old = {}
        if os.path.exists(tp):
            old = json.load(open(tp))
        for k, v in traffic.items():
            v["hbm_bytes"] = v.get("read_bytes", 0.0) + v.get("write_bytes_uncalibrated", 0.0)
            v["note"] = "per launch; read side = FETCH_SIZE KiB x1024 x2 (gfx950 correction)"
            old[k] = v
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from kuiperllama_amd.build import kernel_sources_sha1
        old["_meta"] = {"commit": a.commit or None, "command": a.command, "round": a.round,
                        "kernel_sources_sha1": kernel_sources_sha1(),
                        "note": "HBM bytes per launch from separate rocprofv3 PMC passes; bench.py reads "
                                "roofline.traffic from this file (it is NOT measured inside a bench run)"}
        json.dump(old, open(tp, "w"), indent=1, sort_keys=True)
        print("wrote", tp)


if __name__ == "__main__":
    main()
