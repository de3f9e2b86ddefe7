"""Create / use / destroy cycles of small models in one process - the pattern of the GPU suite, where
kh_model_destroy crashed once in five suite runs of round 6.  Run under tools/dbg/libsegv_bt.so for a native
backtrace.  usage: python tools/stress_destroy.py [seconds] ; KH_KV_VMM=0 selects the plain KV allocation."""
import dataclasses
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
specs = [
    binfmt.ModelSpec(512, 1536, 3, 8, 2, 640, 1024, True, binfmt.FAMILY_LLAMA, False, 64, binfmt.ROPE_HALF, 500000.0,
                     1e-5, "st-gqa"),
    binfmt.ModelSpec(512, 1408, 2, 4, 4, 500, 1024, False, binfmt.FAMILY_LLAMA, True, 64, binfmt.ROPE_INTERLEAVED,
                     10000.0, 1e-5, "st-int8"),
    binfmt.ModelSpec(448, 1216, 2, 7, 1, 700, 4096, True, binfmt.FAMILY_QWEN2, False, 64, binfmt.ROPE_HALF, 1000000.0,
                     1e-6, "st-qwen"),
]
imgs = [binfmt.synth_image(s, seed=70 + i, device="cuda:0") for i, s in enumerate(specs)]
torch.cuda.synchronize()
rng = np.random.default_rng(3)
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    k = n % len(specs)
    spec = specs[k]
    toks = [int(t) for t in rng.integers(0, spec.vocab_size, 601)]
    m = KuiperModel.from_device_image(imgs[k], spec)
    mode = n % 4
    if mode == 0:
        m.prefill_gemm(toks[:600], 0)
        m.predict(toks[600], 600, exec="fused")
    elif mode == 1:
        m.generate(toks[:5], 64, exec="graph")
    elif mode == 2:
        m.prefill(toks[:37], 0)
        m.predict(toks[37], 37, exec="fused")
        m.read_kv(0, 0, 38)
    else:
        m.prefill_gemm(toks[:200], 0)
        m.prefill_gemm(toks[200:530], 200)
        m.logits()
    # keep a second model alive across the first one's destruction now and then (the suite does)
    if n % 7 == 0:
        m2 = KuiperModel.from_device_image(imgs[(k + 1) % len(specs)], specs[(k + 1) % len(specs)])
        m.close()
        m2.generate([1, 2, 3], 16, exec="graph")
        m2.close()
    else:
        m.close()
    n += 1
    if n % 1000 == 0:
        print(f"  {n} cycles, {time.time() - t0:.0f} s", flush=True)
print(f"{n} create/use/destroy cycles in {time.time() - t0:.1f} s, KH_KV_VMM={os.environ.get('KH_KV_VMM', '(default on)')} "
      f"KH_KV_VA_POOL={os.environ.get('KH_KV_VA_POOL', '(default on)')}: no crash")
