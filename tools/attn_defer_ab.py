#!/usr/bin/env python3
"""Decode attention at positions that need several time splits: the split partials merged by the wo kernel
(k_wo_comb, the default) against the ticket + last-arriver merge inside the attention launch
(KH_FLAG_ATTN_MERGE_IN_LAUNCH), same box, same model, same cache contents.  Per position: attention launch,
wo launch (back-to-back over all layers, kh_model_profile_kernel) and the whole graph-replayed step.
Run on the GPU box.  usage: tools/attn_defer_ab.py [workload ...]   (KH_ATTN_TLONG etc. apply)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import _ffi, binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

names = sys.argv[1:] or ["llama3.2-1b"]
dev = torch.device("cuda:0")
POS = (63, 255, 256, 511, 1023, 2047, 4094, 4095, 4096, 8191, 16383, 32767, 32768, 131071)
if os.environ.get("AB_POS"):  # e.g. AB_POS=63,127,191,255
    POS = tuple(int(x) for x in os.environ["AB_POS"].split(","))
for name in names:
    spec = binfmt.PRESETS[name]
    img = binfmt.synth_image(spec, seed=1234, device=dev)
    torch.cuda.synchronize()
    poss = [p for p in POS if p < spec.seq_len]
    modes = (("deferred", 0), ("in-launch", _ffi.KH_FLAG_ATTN_MERGE_IN_LAUNCH)) * 2
    if os.environ.get("AB_MODES"):  # e.g. AB_MODES=deferred : one pass of one mode
        modes = tuple(m for m in modes[:2] if m[0] in os.environ["AB_MODES"].split(","))
    for label, flags in modes:
        m = KuiperModel.from_device_image(img, spec, flags=flags)
        gen = torch.Generator(device=dev)
        gen.manual_seed(7)
        top = max(poss) + 1
        for l in range(spec.n_layers):
            for r0 in range(0, top, 16384):
                n = min(16384, top - r0)
                kv = torch.empty((2, n, spec.kv_dim), dtype=torch.float32, device=dev).normal_(0.0, 1.0, generator=gen)
                m.write_kv_device(l, r0, kv[0], kv[1])
        m.generate([1, 263], 16, exec="graph")
        for p in poss:
            a = m.profile_kernel("attn", p, reps=4)
            w = m.profile_kernel("wo", p, reps=4)
            st = sorted(m.time_step(p, 7))[3]
            kvb = 2.0 * (p + 1) * spec.kv_dim * 4
            print(json.dumps({"model": name, "merge": label, "tlong": os.environ.get("KH_ATTN_TLONG", "default"),
                              "pos": p, "attn_us": round(a, 2), "wo_us": round(w, 2), "attn+wo_us": round(a + w, 2),
                              "step_us": round(st, 1), "kv_MB": round(kvb / 1e6, 2)}), flush=True)
        m.close()
        del m
        torch.cuda.empty_cache()
    del img
    torch.cuda.empty_cache()
