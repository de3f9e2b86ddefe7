#!/usr/bin/env python3
"""Prompt-phase time of the MFMA GEMM prefill alone (kh_model_time_prefill, HIP events on the model
stream): ms and prompt tok/s for 128- ... 1024-token prompts at position 0, best of 5 / 3 (GPU box;
KH_LIB selects an experiment build).   usage: tools/prefill_time.py [label] workload..."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "shipped"
dev = torch.device("cuda:0")
for name in sys.argv[2:] or ["llama3.2-1b"]:
    spec = binfmt.PRESETS[name]
    img = binfmt.synth_image(spec, seed=1234, device=dev)
    torch.cuda.synchronize()
    rng = np.random.default_rng(0)
    toks = [int(t) for t in rng.integers(0, spec.vocab_size, 128)]
    m = KuiperModel.from_device_image(img, spec, max_seq_len=min(spec.seq_len, int(os.environ.get("KH_PT_MAXSEQ", "2048"))))
    m.time_prefill(toks, 0, "gemm")
    ms = min(m.time_prefill(toks, 0, "gemm") for _ in range(5))
    row = {"label": label, "workload": name, "ms_128": round(ms, 4), "prompt_tok_s": round(128 / ms * 1e3)}
    for n in [int(v) for v in os.environ.get("KH_PT_SIZES", "256,384,512,640,1024").split(",") if v]:  # longer prompts: weight passes of up to KH_PG_TMAX tokens (KH_PG_CHUNK to A/B)
        if spec.seq_len >= n:
            lp = [int(t) for t in rng.integers(0, spec.vocab_size, n)]
            m.time_prefill(lp, 0, "gemm")
            ms2 = min(m.time_prefill(lp, 0, "gemm") for _ in range(3))
            row[f"prompt_tok_s_{n}"] = round(n / ms2 * 1e3)
    print(json.dumps(row), flush=True)
    m.close()
    del img, m
    torch.cuda.empty_cache()
