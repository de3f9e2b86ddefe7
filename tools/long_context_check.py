import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from kuiperllama_amd import binfmt
from kuiperllama_amd.model import KuiperModel
dev = torch.device("cuda:0")
spec = binfmt.PRESETS["llama3.2-1b"]
img = binfmt.synth_image(spec, seed=1234, device=dev); torch.cuda.synchronize()
res = {}
for tl in ("4096", "0"):
    os.environ["KH_ATTN_TLONG"] = tl
    m = KuiperModel.from_device_image(img, spec, max_seq_len=8192)
    w, ms = m.generate([1, 263], 4400)
    res[tl] = w
    print("TLONG", tl, "4400 steps", round(ms, 1), "ms", round(4400 / ms * 1e3, 1), "tok/s", flush=True)
    # per-position step latency around the switch
    print({p: round(sorted(m.time_step(p, 5))[2], 1) for p in (127, 1023, 4000, 4094, 4095, 4200)})
    m.close()
a, b = res["4096"], res["0"]
diff = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
print("first difference between group path and per-head path:", diff, "distinct tokens:", len(set(a)))
