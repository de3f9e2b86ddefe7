#!/usr/bin/env python3
"""Decode attention at long positions (run on the GPU box): time of kh_mha_decode_f32 per launch
vs the K/V bytes it must read (2*(pos+1)*kv_dim*4), for the BASELINE geometries."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from kuiperllama_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
GEOMS = {"llama3.2-1b": (32, 8, 64, 131072), "llama2-7b": (32, 32, 128, 2048),
         "qwen2.5-0.5b": (14, 2, 64, 32768)}
LAYERS = 4  # rotate over layers so consecutive launches do not hit the same cache lines
for name, (H, KVH, hs, seq) in GEOMS.items():
    kv_dim = KVH * hs
    kv_mul = H // KVH
    g = torch.Generator(device=dev).manual_seed(1)
    k = torch.randn(LAYERS, seq, kv_dim, device=dev, generator=g)
    v = torch.randn(LAYERS, seq, kv_dim, device=dev, generator=g)
    q = torch.randn(H * hs, device=dev, generator=g)
    out = torch.empty(H * hs, device=dev)
    ws = ops.mha_decode_workspace(H, hs, seq, dev)
    p = 127
    while p < seq:
        reps = 8
        for l in range(LAYERS):
            ops.mha_decode(p, H, l, seq, kv_dim, kv_mul, hs, out, q, k, v, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            for l in range(LAYERS):
                ops.mha_decode(p, H, l, seq, kv_dim, kv_mul, hs, out, q, k, v, ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * LAYERS)
        b = 2 * (p + 1) * kv_dim * 4
        print(f"{name:14s} pos {p:7d}  {us:9.2f} us  KV {b / 1e6:9.2f} MB  {b / us / 1e6:7.3f} TB/s",
              flush=True)
        p = p * 4 + 3
    del k, v
