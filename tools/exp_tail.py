#!/usr/bin/env python3
"""Does the last, partly filled round of row pairs cost its full time?  ffn13 of a Llama-2-7B-shaped
model (dim 4096, 8 layers) with hidden sizes that do / do not divide into whole rounds of the
launch: us per launch and achieved bytes/s.
  python tools/exp_tail.py [int8|f32]"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

quant = (sys.argv[1] if len(sys.argv) > 1 else "int8") == "int8"
base = binfmt.PRESETS["llama2-7b-int8" if quant else "llama2-7b"]
for hidden in (10240, 11008, 11264, 12288, 14336, 16384):
    spec = dataclasses.replace(base, hidden_dim=hidden, n_layers=6, name=f"h{hidden}")
    img = binfmt.synth_image(spec, seed=1, device=torch.device("cuda:0"))
    torch.cuda.synchronize()
    m = KuiperModel.from_device_image(img, spec, max_seq_len=256)
    m.generate([1, 263], 16, exec="graph")
    k = m.profile_kernels(8, reps=20)
    nbytes = 2 * hidden * spec.dim * (1 if quant else 4) * (1 + (4 / 64 if quant else 0))
    print(f"hidden {hidden:6d} pairs/CU {hidden / 256:7.3f}  ffn13 {k['ffn13']:7.3f} us  {nbytes / k['ffn13'] / 1e6:6.2f} TB/s   "
          f"w2 {k['w2']:7.3f} us", flush=True)
    m.close()
    del img
    torch.cuda.empty_cache()
