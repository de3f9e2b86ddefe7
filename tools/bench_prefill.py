#!/usr/bin/env python3
"""Prompt-phase throughput (run on the GPU box): generate() with a P-token prompt and P total
steps, with the prefill (4 prompt tokens per weight pass, kh_prefill.h) and with the reference's
one-token-per-step prompt phase (KH_PREFILL=0).  HIP-event time of the whole loop."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["llama3.2-1b", "llama2-7b-int8"]:
    spec = binfmt.PRESETS[name]
    img = binfmt.synth_image(spec, seed=1234, device=dev)
    torch.cuda.synchronize()
    rng = np.random.default_rng(0)
    for P in (33, 129, 513):
        prompt = [int(t) for t in rng.integers(0, spec.vocab_size, P)]
        res = {}
        for mode in ("1", "0"):
            os.environ["KH_PREFILL"] = mode
            m = KuiperModel.from_device_image(img, spec, max_seq_len=1024)
            m.generate(prompt, P)  # warm (graph capture, buffers)
            best = min(m.generate(prompt, P)[1] for _ in range(3))
            w = m.generate(prompt, P)[0]
            res[mode] = (best, w[-1])
            m.close()
        os.environ.pop("KH_PREFILL", None)
        assert res["1"][1] == res["0"][1]
        print(f"{name:16s} prompt {P - 1:4d}+1 tokens: prefill {res['1'][0]:8.2f} ms "
              f"({(P - 1) / res['1'][0] * 1e3:8.0f} prompt tok/s)   token-by-token {res['0'][0]:8.2f} ms "
              f"({(P - 1) / res['0'][0] * 1e3:8.0f} tok/s)   x{res['0'][0] / res['1'][0]:.2f}", flush=True)
    del img
    torch.cuda.empty_cache()
