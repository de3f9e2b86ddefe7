#!/usr/bin/env python3
"""HBM traffic of the decode-attention launch deep in the context (Llama-3.2-1B, full 131072-row cache): run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -o b -- python tools/pmc_attn_long.py <pos>
(one position per invocation: the dispatches of a pass are then all that position), then
  python tools/pmc_attn_long.py --report <dir> <pos>
prints read bytes per launch (FETCH_SIZE KiB x 1024 x 2: the gfx950 half-count correction of
MI355X_MICROARCH.md) against the algorithmic K/V bytes 2 * (pos + 1) * kv_dim * 4."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if sys.argv[1] == "--report":
    d, pos = sys.argv[2], int(sys.argv[3])
    db = sorted(glob.glob(os.path.join(d, "**", "*results.db"), recursive=True))[0]
    con = sqlite3.connect(db)
    for name, n, avg, mn, mx in con.execute(
            "select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
            "where counter_name = 'FETCH_SIZE' and kernel_name like '%k_attn_decode%' group by kernel_name"):
        algo = 2.0 * (pos + 1) * 512 * 4
        rd = avg * 1024 * 2
        print(f"pos {pos}: {name.split('(')[0]}  {n} launches  read {rd / 1e6:.2f} MB per launch "
              f"(FETCH_SIZE {avg:.0f} KiB, min {mn:.0f}, max {mx:.0f})  algorithmic K/V {algo / 1e6:.2f} MB  "
              f"ratio {rd / algo:.3f}")
    sys.exit(0)

import torch  # noqa: E402

from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

pos = int(sys.argv[1])
spec = binfmt.PRESETS["llama3.2-1b"]
dev = torch.device("cuda:0")
img = binfmt.synth_image(spec, seed=1234, device=dev)
m = KuiperModel.from_device_image(img, spec)
gen = torch.Generator(device=dev)
gen.manual_seed(7)
for layer in range(spec.n_layers):
    for r0 in range(0, pos + 1, 16384):
        n = min(16384, pos + 1 - r0)
        kv = torch.empty((2, n, spec.kv_dim), dtype=torch.float32, device=dev).normal_(0.0, 1.0, generator=gen)
        m.write_kv_device(layer, r0, kv[0], kv[1])
torch.cuda.synchronize()
us = m.profile_kernel("attn", pos, reps=2)  # back to back over all layers, twice
print(f"pos {pos}: attention launch {us:.2f} us per layer", flush=True)
m.close()
