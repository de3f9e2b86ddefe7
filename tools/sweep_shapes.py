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


KEY = {"W2": "w2", "WO": "wo", "QKV": "qkv", "FFN": "ffn13", "CLS": "cls", "ATTN": "attn"}


def run(tag, env):
    for k in ("KH_SHAPE_QKV", "KH_SHAPE_WO", "KH_SHAPE_W2", "KH_SHAPE_FFN", "KH_SHAPE_CLS",
              "KH_ATTN_WG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = KuiperModel.from_device_image(img, spec, max_seq_len=512)
    m.generate([1, 263], 16)
    best = 1e9
    for _ in range(2):
        _, ms = m.generate([1, 263], 128)
        best = min(best, ms)
    prof = m.profile_kernels(64, 8)  # back-to-back launches, no event overhead
    m.close()
    row = {"tag": tag, "tok_s": round(128e3 / best, 1), **{k: round(v, 2) for k, v in prof.items()}}
    res.append(row)
    print(json.dumps(row), flush=True)


classes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["W2", "WO", "QKV"]
run("default", {})
us = (2, 4) if spec.quant else (4, 8)
for cls in classes:
    if cls == "ATTN":
        for wg in (256, 512):
            run(f"ATTN wg{wg}", {"KH_ATTN_WG": str(wg)})
        continue
    maxsplit = {"QKV": 2, "FFN": 1, "CLS": 1}.get(cls, 4)
    wgs = (256, 512)
    grids = (256, 384, 512, 768, 1024)
    for wg, sp, u, g in itertools.product(wgs, (1, 2, 4), us, grids):
        if sp > maxsplit or g * wg > 1024 * 256:
            continue
        run(f"{cls} wg{wg} s{sp} u{u} g{g}", {f"KH_SHAPE_{cls}": f"{sp},{u},{g},{wg}"})
best = {}
for r in res:
    c = r["tag"].split()[0]
    k = KEY.get(c)
    if k and (c not in best or r[k] < best[c][k]):
        best[c] = r
print("BEST", json.dumps(best))
