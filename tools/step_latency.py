#!/usr/bin/env python3
"""Single-step latency (us, median of 7 graph replays) at a list of positions (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from kuiperllama_amd import binfmt
from kuiperllama_amd.model import KuiperModel
name = sys.argv[1] if len(sys.argv) > 1 else "llama3.2-1b"
poss = [int(p) for p in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,64,127,255,511,1023,2047,4000".split(","))]
spec = binfmt.PRESETS[name]
img = binfmt.synth_image(spec, seed=1234, device=torch.device("cuda:0")); torch.cuda.synchronize()
m = KuiperModel.from_device_image(img, spec, max_seq_len=min(spec.seq_len, 8192))
n = min(max(poss) + 2, m.cfg.cache_len)
m.generate([1, 263], n)
print(name, {p: round(sorted(m.time_step(p, 7))[3], 1) for p in poss if p < n}, flush=True)
