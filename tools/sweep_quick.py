#!/usr/bin/env python3
"""Targeted launch-shape A/B for the decode GEMV kernels (run on the GPU box): a short explicit
candidate list per kernel class ("split,u,grid,wg") instead of tools/sweep_shapes.py's full
product.  Prints one JSON row per candidate: tok/s of a 128-step greedy run and the back-to-back
per-kernel us.   usage: tools/sweep_quick.py <workload> CLASS=s,u,g,wg[;s,u,g,wg...] ..."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

workload = sys.argv[1]
spec = binfmt.PRESETS[workload]
dev = torch.device("cuda:0")
img = binfmt.synth_image(spec, seed=1234, device=dev)
torch.cuda.synchronize()
KEY = {"W2": "w2", "WO": "wo", "QKV": "qkv", "FFN": "ffn13", "CLS": "cls"}


def run(tag, env):
    for k in list(os.environ):
        if k.startswith("KH_SHAPE_"):
            del os.environ[k]
    os.environ.update(env)
    m = KuiperModel.from_device_image(img, spec, max_seq_len=512)
    m.generate([1, 263], 16)
    ms = min(m.generate([1, 263], 128)[1] for _ in range(3))
    prof = m.profile_kernels(64, 8)
    m.close()
    row = {"tag": tag, "tok_s": round(128e3 / ms, 1), **{k: round(v, 2) for k, v in prof.items()}}
    print(json.dumps(row), flush=True)
    return row


base = run("default", {})
for arg in sys.argv[2:]:
    cls, cands = arg.split("=")
    for c in cands.split(";"):
        r = run(f"{cls} {c}", {f"KH_SHAPE_{cls}": c})
        k = KEY[cls]
        print(f"   {cls} {c}: {k} {base[k]} -> {r[k]} us, tok/s {base['tok_s']} -> {r['tok_s']}", flush=True)
