#!/usr/bin/env python3
"""kh_model_create_from_file on an image in /dev/shm: whole-call and upload-only GB/s.
Loads 1-3 back to back (each right after the previous model's hipFree of weights + KV cache), load 4 after a pause:
is a slow create the allocator handing back memory that is still being released?
usage: tools/load_probe.py [workload]   (run on the GPU box)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama3.2-1b"
spec = binfmt.PRESETS[name]
img = binfmt.synth_image(spec, seed=1234, device=torch.device("cuda:0"))
path = f"/dev/shm/kh_probe_{os.getpid()}.bin"
img.cpu().numpy().tofile(path)
n = int(img.numel())
del img
torch.cuda.empty_cache()
try:
    for i in range(4):
        if i == 3:
            time.sleep(3.0)
        t0 = time.perf_counter()
        m = KuiperModel.from_file(path, spec)
        wall = time.perf_counter() - t0
        up = m.load_ms
        kv0 = m.kv_bytes()
        t1 = time.perf_counter()
        words, _ = m.generate([1, 263], 8)
        first_gen_ms = (time.perf_counter() - t1) * 1e3  # includes mapping the first KV chunks + graph capture
        kv1 = m.kv_bytes()
        m.close()
        print(json.dumps({"load": i + 1, "after": "a 3-s pause" if i == 3 else ("the image's hipFree" if i == 0 else "the previous model's hipFree"),
                          "workload": name, "GB": round(n / 1e9, 2), "create_ms": round(wall * 1e3, 1),
                          "GB/s": round(n / wall / 1e9, 1), "upload_ms": round(up, 1),
                          "upload_GB/s": round(n / up / 1e6, 1), "other_ms": round(wall * 1e3 - up, 1),
                          "kv_reserved_GB": round(kv0[0] / 1e9, 2), "kv_committed_MB_after_create": round(kv0[1] / 1e6, 1),
                          "kv_committed_MB_after_8_steps": round(kv1[1] / 1e6, 1), "first_generate_ms": round(first_gen_ms, 1),
                          "words": words[:4]}), flush=True)
finally:
    os.unlink(path)
