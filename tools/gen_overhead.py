#!/usr/bin/env python3
"""Host-side overhead of one generate() call: wall clock (perf_counter around the call + device sync) minus the
HIP-event time of the step loop, for run lengths that exercise the 1 / 2 / 4 / 8-step graphs.  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kuiperllama_amd import binfmt  # noqa: E402
from kuiperllama_amd.model import KuiperModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama3.2-1b"
spec = binfmt.PRESETS[name]
dev = torch.device("cuda:0")
img = binfmt.synth_image(spec, seed=1234, device=dev)
m = KuiperModel.from_device_image(img, spec)
m.generate([1, 263], 40)
for steps in (3, 4, 5, 8, 9, 16, 17, 20, 24, 32, 128):
    best = None
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, ev_ms = m.generate([1, 263], steps)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        if best is None or wall_ms < best[0]:
            best = (wall_ms, ev_ms)
    print(f"steps {steps:4d}: wall {best[0]:8.3f} ms  events {best[1]:8.3f} ms  overhead {1e3 * (best[0] - best[1]):7.1f} us"
          f"  -> {steps / best[0] * 1e3:7.1f} tok/s by wall, {steps / best[1] * 1e3:7.1f} by events", flush=True)

# the bench's own sequence: a NEW model, a 5-step warm-up, then ONE timed 20-step run (its 8-step graph has been
# captured by the warm-up call but never launched)
m.close()
for rep in range(4):
    m = KuiperModel.from_device_image(img, spec)
    m.generate([1, 263], 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, ev_ms = m.generate([1, 263], 20)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    print(f"fresh model, warm-up 5, first 20-step run: wall {wall_ms:8.3f} ms events {ev_ms:8.3f} ms -> "
          f"{20 / wall_ms * 1e3:7.1f} tok/s by wall, {20 / ev_ms * 1e3:7.1f} by events", flush=True)
    m.close()

# and with a second timed run right behind the first (is it the first LAUNCH of the big graph or the state of the GPU?)
for rep in range(3):
    m = KuiperModel.from_device_image(img, spec)
    m.generate([1, 263], 5)
    out = []
    for k in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, ev_ms = m.generate([1, 263], 20)
        torch.cuda.synchronize()
        out.append(20 / ((time.perf_counter() - t0)) )
    print("fresh model, warm-up 5, three 20-step runs in a row: " + " ".join(f"{x:7.1f}" for x in out), flush=True)
    m.close()
