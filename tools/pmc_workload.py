#!/usr/bin/env python3
"""Minimal workload for counter / kernel-trace passes: build one model, run a few graph-replayed
decode steps and (optionally) GEMM / GEMV prefills of 128 tokens.  Nothing else (no CPU baseline, no
per-kernel event timing), so that a profiler pass sees only the kernels of the hot path.
  python tools/pmc_workload.py <workload> [--steps N] [--prefill gemm,gemv] [--reps R]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--prefill", default="")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--prompt", type=int, default=128, help="prompt tokens per timed prefill")
ap.add_argument("--pos0", type=int, default=0, help="position of the first prompt token")
a = ap.parse_args()
spec = binfmt.PRESETS[a.workload]
img = binfmt.synth_image(spec, seed=1234, device=torch.device("cuda:0"))
torch.cuda.synchronize()
m = KuiperModel.from_device_image(img, spec, max_seq_len=0)
if a.steps > 0:
    for _ in range(a.reps):
        m.generate([1, 263], a.steps, exec="graph")
rng = np.random.default_rng(0)
pp = [int(t) for t in rng.integers(0, spec.vocab_size, a.prompt)]
for mode in [x for x in a.prefill.split(",") if x]:
    for _ in range(a.reps):
        ms = m.time_prefill(pp, a.pos0, mode)
    at = f" at pos {a.pos0}" if a.pos0 else ""
    print(f"prefill {mode}: {ms:.3f} ms for {a.prompt} tokens{at} = {a.prompt * 1e3 / ms:.0f} tok/s", flush=True)
m.close()
