#!/usr/bin/env python3
"""Generate the per-round kernel table of profiles/README.md FROM the committed profile files, so that no
microsecond quoted there can drift from the CSV it came from.

    python tools/profiles_readme.py [--round r6] [--write]

Sources: profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench command, our kernels),
profiles/pmc_traffic.json (separate --pmc passes: HBM bytes per launch), profiles/<round>_bench.json and
<round>_bench_steps20.json (the bench lines).  The algorithmic bytes per launch come from
kuiperllama_amd.binfmt.ModelSpec.kernel_bytes().  --write replaces the block between
`<!-- generated:<round> begin -->` and `<!-- generated:<round> end -->` in profiles/README.md (appends it at the
top of the round's section when absent); without it the block is printed.  tests/test_bench_logic.py runs the
generator on the committed files and requires README.md to contain exactly its output."""
import argparse
import csv
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
PEAK = 8e12


def classify(name: str, wg: int):
    """-> (workload, kernel class) of a decode GEMV instantiation of the bench run, or None."""
    m = re.match(r"k_(qkv|ffn13_ring|ffn13|gemv_res|cls_ring|cls)<(true|false|\d+)", name)
    if not m:
        return None
    k, first = m.group(1), m.group(2)
    if k.endswith("_ring"):
        return "llama2-7b-int8", k[:-5]
    quant = first == "true"
    wl = "llama2-7b-int8" if quant else "llama3.2-1b"
    if k == "gemv_res":
        return wl, ("w2" if wg == 512 else "wo")
    return wl, k


def block(rnd: str) -> str:
    from kuiperllama_amd import binfmt
    prof = os.path.join(ROOT, "profiles")
    rows = []
    with open(os.path.join(prof, f"{rnd}_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            c = classify(r["kernel"], int(r["wg"]))
            if c and int(r["calls"]) >= 1000:  # the decode launches of the two metric workloads (prefill twins: k_pf_*)
                rows.append((c, r))
    # HBM traffic per launch of every kernel INSTANTIATION from the round's own counter passes
    # (<round>_pmc_<workload>.csv: FETCH_SIZE / WRITE_SIZE in KiB; read side x2 = the gfx950 half-count correction of
    # a 16 B/lane stream, MI355X_MICROARCH.md HBM section - the same formula as pmc_traffic.json)
    traffic = {}
    for wl in ("llama3.2-1b", "llama2-7b-int8"):
        try:
            with open(os.path.join(prof, f"{rnd}_pmc_{wl}.csv")) as f:
                acc = {}
                for r in csv.DictReader(f):
                    acc.setdefault(r["kernel"], {})[r["counter"]] = float(r["avg_kib"])
            for k, v in acc.items():
                if "FETCH_SIZE" in v:
                    traffic[(wl, k)] = v["FETCH_SIZE"] * 1024 * 2 + v.get("WRITE_SIZE", 0.0) * 1024
        except OSError:
            pass
    out = [f"<!-- generated:{rnd} begin (tools/profiles_readme.py --round {rnd} --write; do not edit by hand) -->",
           f"Decode GEMV launches of the two metric workloads in `{rnd}_kernel_stats.csv` (rocprofv3 `--kernel-trace --stats` "
           f"of the bench command); bytes = `ModelSpec.kernel_bytes()`; HBM traffic = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 "
           f"of the same instantiation in `{rnd}_pmc_<workload>.csv` (separate `--pmc` passes).",
           "",
           "| workload | kernel (instantiation) | calls | avg µs | bytes / launch | TB/s | of 8 TB/s | HBM traffic / algorithmic |",
           "|---|---|---|---|---|---|---|---|"]
    order = {"qkv": 0, "wo": 1, "ffn13": 2, "w2": 3, "cls": 4}
    for (wl, k), r in sorted(rows, key=lambda x: (x[0][0] != "llama3.2-1b", order[x[0][1]])):
        b = binfmt.PRESETS[wl].kernel_bytes()[k]
        us = float(r["avg_us"])
        tbs = b / (us * 1e-6) / 1e12
        tr = traffic.get((wl, r["kernel"]))
        ratio = f"{tr / b:.3f}" if tr else "—"
        out.append(f"| {wl} | {k} `{r['kernel']}` | {int(r['calls'])} | {us:.2f} | {b / 1e6:.1f} MB | {tbs:.2f} | "
                   f"{b / (us * 1e-6) / PEAK:.3f} | {ratio} |")
    for tag, fn in (("default command", f"{rnd}_bench.json"), ("driver form `--steps 20 --warmup 5`", f"{rnd}_bench_steps20.json")):
        p = os.path.join(prof, fn)
        if not os.path.exists(p):
            continue
        try:
            d = json.loads(open(p).read().strip().split("\n")[-1])
        except (ValueError, IndexError):
            continue
        sec = d.get("secondary") or {}
        rf, srf = d.get("roofline", {}), sec.get("roofline", {})
        out.append("")
        out.append(f"`{fn}` ({tag}): Llama-3.2-1B fp32 **{d['value']:.1f} tok/s** ({d['ms_per_step']:.4f} ms/step, step "
                   f"{rf.get('step', {}).get('frac', float('nan')):.3f} of 8 TB/s, ffn13 {rf.get('avg_launch_us', float('nan')):.2f} µs = "
                   f"{rf.get('frac', float('nan')):.3f}); Llama-2-7B int8 **{sec.get('value', float('nan')):.1f} tok/s** (step "
                   f"{srf.get('step', {}).get('frac', float('nan')):.3f}, ffn13 {srf.get('avg_launch_us', float('nan')):.2f} µs = "
                   f"{srf.get('frac', float('nan')):.3f}); 128-step figures {d.get('tok_s_128_steps', float('nan')):.1f} / "
                   f"{sec.get('tok_s_128_steps', float('nan')):.1f}.")
    out.append(f"<!-- generated:{rnd} end -->")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r6")
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    b = block(a.round)
    if not a.write:
        print(b)
        return
    p = os.path.join(ROOT, "profiles", "README.md")
    s = open(p).read()
    pat = re.compile(rf"<!-- generated:{a.round} begin.*?<!-- generated:{a.round} end -->", re.S)
    if pat.search(s):
        s = pat.sub(lambda _: b, s)
    else:
        head = re.search(rf"^## Round {a.round[1:]}\b.*$", s, re.M)
        if not head:
            raise SystemExit(f"profiles/README.md has no '## Round {a.round[1:]}' section")
        s = s[:head.end()] + "\n\n" + b + "\n" + s[head.end():]
    open(p, "w").write(s)
    print(f"profiles/README.md: block of {a.round} written")


if __name__ == "__main__":
    main()
