#!/usr/bin/env python3
"""Launch-shape sweep for the small GEMV kernels (run on the GPU box).  For each candidate
(split, u, grid) of one kernel class, rebuild the model with KH_SHAPE_<K> set and report the
HIP-event average duration of that class (kh_model_profile_step) and whole-step tokens/s."""
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "llama3.2-1b"
spec = binfmt.PRESETS[workload]
dev = torch.device("cuda:0")
img = binfmt.synth_image(spec, seed=1234, device=dev)
torch.cuda.synchronize()
res = []


def run(tag, env):
    for k in ("KH_SHAPE_QKV", "KH_SHAPE_WO", "KH_SHAPE_W2", "KH_SHAPE_FFN", "KH_SHAPE_CLS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = KuiperModel.from_device_image(img, spec, max_seq_len=512)
    m.generate([1, 263], 16)
    best = 1e9
    for _ in range(3):
        _, ms = m.generate([1, 263], 128)
        best = min(best, ms)
    prof = m.profile_step(64, 8)
    m.close()
    row = {"tag": tag, "env": env, "tok_s": 128e3 / best,
           **{k: round(v["avg_us"], 2) for k, v in prof.items()}}
    res.append(row)
    print(json.dumps(row), flush=True)


run("default", {})
us = (2, 4) if spec.quant else (4, 8)
for cls in ("W2", "WO", "QKV"):
    maxsplit = 2 if cls == "QKV" else 4
    for sp, u, g in itertools.product((1, 2, 4), us, (256, 512, 768, 1024)):
        if sp > maxsplit:
            continue
        run(f"{cls} s{sp} u{u} g{g}", {f"KH_SHAPE_{cls}": f"{sp},{u},{g}"})
