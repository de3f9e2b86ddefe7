#!/usr/bin/env python3
"""Per-kernel view of the GEMM prefill at several chunk sizes.

  run   (under rocprofv3 --kernel-trace --output-format csv):  prefill_kernels.py run <workload> T...
  table (on the *_kernel_trace.csv it leaves):                 prefill_kernels.py table <csv>

`run` times one prefill of T tokens per listed T (after a warm-up of each); `table` groups the dispatches by
(kernel, grid, workgroup, LDS) - chunk sizes differ in grid, so every T gets its own rows - and prints
launches, average us and the share of the total."""
import csv
import os
import re
import sys
from collections import OrderedDict


def run(name, sizes):
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from kuiperllama_amd import binfmt
    from kuiperllama_amd.model import KuiperModel
    spec = binfmt.PRESETS[name]
    img = binfmt.synth_image(spec, seed=1234, device=torch.device("cuda:0"))
    torch.cuda.synchronize()
    m = KuiperModel.from_device_image(img, spec, max_seq_len=min(spec.seq_len, 2048))
    rng = np.random.default_rng(0)
    for n in sizes:
        toks = [int(t) for t in rng.integers(0, spec.vocab_size, n)]
        m.time_prefill(toks, 0, "gemm")
        ms = min(m.time_prefill(toks, 0, "gemm") for _ in range(3))
        print(f"{name} T={n}: {ms:.3f} ms = {n / ms * 1e3:.0f} prompt tok/s", flush=True)
    m.close()


def table(path):
    rows = OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            nm = re.sub(r"^void ", "", r["Kernel_Name"])
            nm = re.sub(r"\(.*$", "", nm)
            if not nm.startswith(("k_pg_", "k_pf_", "k_emb")):
                continue
            key = (nm, f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}', r["Workgroup_Size_X"], r["LDS_Block_Size"])
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n, s = rows.get(key, (0, 0.0))
            rows[key] = (n + 1, s + d)
    print(f'{"kernel":58s} {"grid(thr)":>14s} {"wg":>4s} {"lds":>6s} {"n":>5s} {"avg_us":>8s}')
    for (nm, grid, wg, lds), (n, s) in rows.items():
        print(f"{nm:58s} {grid:>14s} {wg:>4s} {lds:>6s} {n:5d} {s / n:8.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], [int(a) for a in sys.argv[3:]])
    else:
        table(sys.argv[2])
