#!/usr/bin/env python3
"""tok/s (128 greedy steps, hipGraph, best of 3) and back-to-back per-kernel us of one workload.
Run on the GPU box; KH_LIB selects an experiment build, KH_SHAPE_* a launch shape.
usage: tools/kprof.py <workload> [label]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7b-int8"
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get("KH_LIB", "shipped"))
spec = binfmt.PRESETS[name]
dev = torch.device("cuda:0")
img = binfmt.synth_image(spec, seed=1234, device=dev)
torch.cuda.synchronize()
m = KuiperModel.from_device_image(img, spec, max_seq_len=min(spec.seq_len, 4096))
m.generate([1, 263], 16)
ms = min(m.generate([1, 263], 128)[1] for _ in range(3))
words = m.generate([1, 263], 128)[0]
k = m.profile_kernels(64, reps=8)
env = {e: os.environ[e] for e in os.environ if e.startswith("KH_") and e != "KH_LIB"}
print(json.dumps({"label": label, "workload": name, "tok_s": round(128 / (ms * 1e-3), 1),
                  "kernels_us": {a: round(b, 2) for a, b in k.items()},
                  "sum_us": round(sum(v * (spec.n_layers if a not in ("cls", "sample") else 1) for a, v in k.items()), 1),
                  "env": env, "words_crc": hash(tuple(words)) & 0xffffffff}), flush=True)
