#!/usr/bin/env python3
"""Decode attention launch time per layer at deep positions for several switch points of the GQA
group path (KH_ATTN_TLONG: pos + 1 from which one workgroup per (kv group, split) replaces one per
(head, split); 0 = never).  Run on the GPU box.  usage: tools/attn_tlong_sweep.py [workload]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama3.2-1b"
spec = binfmt.PRESETS[name]
dev = torch.device("cuda:0")
img = binfmt.synth_image(spec, seed=1234, device=dev)
torch.cuda.synchronize()
poss = [p for p in (1023, 2047, 4095, 8191, 16383, 32767, 65535, 131071) if p < spec.seq_len]
for tl in os.environ.get("KH_SWEEP_TLONGS", "0,2048,4096,8192,16384,32768").split(","):
    os.environ["KH_ATTN_TLONG"] = tl
    m = KuiperModel.from_device_image(img, spec)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    top = max(poss) + 1
    for l in range(spec.n_layers):
        for r0 in range(0, top, 16384):
            n = min(16384, top - r0)
            kv = torch.empty((2, n, spec.kv_dim), dtype=torch.float32, device=dev).normal_(0.0, 1.0, generator=gen)
            m.write_kv_device(l, r0, kv[0], kv[1])
    row = {"tlong": int(tl)}
    for p in poss:
        us = m.profile_kernel("attn", p, reps=4)
        row[str(p)] = round(us, 2)
    print(json.dumps(row), flush=True)
    m.close()
    del m
    torch.cuda.empty_cache()
