#!/usr/bin/env python3
"""Prompt prefill deep in the context: ms for a 512-token (and 128-token) pass of the GEMM prefill at start
positions 0 / 4096 / 16384 / 32256 (kh_model_time_prefill; the cache rows before the start position hold
zeros - timing only).  KH_PG_ATTN_QT=1 is the one-query-tile attention kernel.  usage: pattn_time.py [label] [workload...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "shipped"
for name in sys.argv[2:] or ["llama3.2-1b"]:
    spec = binfmt.PRESETS[name]
    img = binfmt.synth_image(spec, seed=1234, device=torch.device("cuda:0"))
    torch.cuda.synchronize()
    cap = min(spec.seq_len, 32768)
    m = KuiperModel.from_device_image(img, spec, max_seq_len=cap)
    rng = np.random.default_rng(0)
    row = {"label": label, "workload": name}
    for n in (512, 128):
        toks = [int(t) for t in rng.integers(0, spec.vocab_size, n)]
        for pos0 in (0, 4096, 16384, 32256):
            if pos0 + n > cap:
                continue
            m.time_prefill(toks, pos0, "gemm")
            ms = min(m.time_prefill(toks, pos0, "gemm") for _ in range(3))
            row[f"ms_{n}_at_{pos0}"] = round(ms, 3)
    print(json.dumps(row), flush=True)
    m.close()
    del img, m
    torch.cuda.empty_cache()
