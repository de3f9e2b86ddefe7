#!/usr/bin/env python3
"""Full-size HuggingFace pins for BASELINE configs 2 and 4 ("token-for-token vs hf_infer/llama3_infer.py /
qwen2_infer.py", /root/reference/hf_infer/llama3_infer.py:1-18).

The reference checks its Llama-3 / Qwen2 builds against `AutoModelForCausalLM` of the same checkpoint.  There is no
network here, so the checkpoint is a RANDOM-INIT `LlamaForCausalLM` / `Qwen2ForCausalLM` at the full Llama-3.2-1B /
Qwen2.5-0.5B geometry built from a fixed seed (same docker image here and on the GPU box => same torch => same
bytes).  A 5 GB / 2 GB image cannot be committed; what is committed (tests/golden/hf_full_<name>.npz) is
    tokens   the fed token ids
    logits   HF's own logits at four positions of that sequence (float32 [4, vocab], ~2 MB)
    sha256   of the .bin image the REFERENCE exporter (tools/export.py / export_qwen2.py: load_hf_model +
             legacy_export) wrote for that model in the build container
and the GPU test (tests/test_hf_fullsize_gpu.py) rebuilds the model from the seed, writes the image with
`export_image` below - this repo's restatement of that export path, asserted byte-identical to the reference
exporter's file when the fixture is made - checks the sha256, and compares the HIP logits with HF's.

`python tests/golden/hf_fullsize.py` (build container only: needs /root/reference) regenerates the fixtures.
"""
import hashlib
import os
import struct
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from kuiperllama_amd import binfmt  # noqa: E402

SEQ_LEN = 256  # max_position_embeddings of the fixture: the header's seq_len and the rows of the exported freqs tables
T = 6          # fed tokens
KEEP = (1, 2, 4, 5)  # positions whose HF logits are committed

CASES = {
    "llama3.2-1b": dict(kind="llama", seed=2101, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                        num_attention_heads=32, num_key_value_heads=8, vocab_size=128256, rope_theta=500000.0,
                        rms_norm_eps=1e-5),
    "qwen2.5-0.5b": dict(kind="qwen2", seed=2102, hidden_size=896, intermediate_size=4864, num_hidden_layers=24,
                         num_attention_heads=14, num_key_value_heads=2, vocab_size=151936, rope_theta=1000000.0,
                         rms_norm_eps=1e-6),
}


def spec_of(name: str) -> binfmt.ModelSpec:
    c = CASES[name]
    fam = binfmt.FAMILY_LLAMA if c["kind"] == "llama" else binfmt.FAMILY_QWEN2
    return binfmt.ModelSpec(c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"], c["num_attention_heads"],
                            c["num_key_value_heads"], c["vocab_size"], SEQ_LEN, True, fam, False, 64, binfmt.ROPE_HALF,
                            c["rope_theta"], c["rms_norm_eps"], "hf-full-" + name)


def build_hf(name: str):
    """Random-init HF model of the BASELINE geometry + the fed tokens, deterministic in the seed."""
    import transformers
    c = dict(CASES[name])
    kind, seed = c.pop("kind"), c.pop("seed")
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    common = dict(c, max_position_embeddings=SEQ_LEN, tie_word_embeddings=True, initializer_range=0.02,
                  attention_dropout=0.0)
    if kind == "llama":
        cfg = transformers.LlamaConfig(**common, rope_scaling=None, attention_bias=False, mlp_bias=False)
        hf = transformers.LlamaForCausalLM(cfg)
    else:
        cfg = transformers.Qwen2Config(**common, use_sliding_window=False)
        hf = transformers.Qwen2ForCausalLM(cfg)
    hf = hf.eval().float()
    with torch.no_grad():
        for n, p_ in hf.named_parameters():
            if n.endswith("norm.weight"):
                p_.copy_(1.0 + 0.1 * torch.randn(p_.shape, generator=gen))
            if n.endswith(".bias"):
                p_.copy_(0.02 * torch.randn(p_.shape, generator=gen))
    tokens = torch.randint(0, c["vocab_size"], (T,), generator=gen).tolist()
    return hf, tokens


def export_image(hf, name: str) -> np.ndarray:
    """The bytes tools/export.py::legacy_export (:78-131; export_qwen2.py adds the q / k / v bias after each
    projection's weights) writes for the model its load_hf_model (:530-590) builds from an HF checkpoint: header of 7
    int32, then fp32 tensors in the legacy llama2.c order - HF's q / k rows as they are (load_hf_model defines
    permute_reverse and never calls it: the rotate-half layout is kept, which is why the C++ needs LLAMA3_SUPPORT /
    QWEN2_SUPPORT) - the reference Transformer's own freqs_cos / freqs_sin tables (theta 10000 whatever the model's
    rope_theta is: model.py:41-47, 227; the loaders skip them) and no classifier for a tied model."""
    c = CASES[name]
    sd = hf.state_dict()
    L, d, heads = c["num_hidden_layers"], c["hidden_size"], c["num_attention_heads"]
    qwen = c["kind"] == "qwen2"
    assert torch.equal(sd["model.embed_tokens.weight"], sd["lm_head.weight"])  # tied: positive vocab, no wcls
    parts = [np.frombuffer(struct.pack("iiiiiii", d, c["intermediate_size"], L, heads, c["num_key_value_heads"],
                                       c["vocab_size"], SEQ_LEN), dtype=np.uint8)]

    def put(t):
        parts.append(t.detach().to(torch.float32).contiguous().view(-1).numpy().view(np.uint8))

    put(sd["model.embed_tokens.weight"])
    for i in range(L):
        put(sd[f"model.layers.{i}.input_layernorm.weight"])
    for proj in ("q_proj", "k_proj", "v_proj"):
        for i in range(L):
            put(sd[f"model.layers.{i}.self_attn.{proj}.weight"])
            if qwen:
                put(sd[f"model.layers.{i}.self_attn.{proj}.bias"])
    for i in range(L):
        put(sd[f"model.layers.{i}.self_attn.o_proj.weight"])
    for i in range(L):
        put(sd[f"model.layers.{i}.post_attention_layernorm.weight"])
    for proj in ("gate_proj", "down_proj", "up_proj"):  # w1, w2, w3
        for i in range(L):
            put(sd[f"model.layers.{i}.mlp.{proj}.weight"])
    put(sd["model.norm.weight"])
    hd = d // heads
    freqs = 1.0 / (10000.0 ** (torch.arange(0, hd, 2)[: (hd // 2)].float() / hd))
    fr = torch.outer(torch.arange(SEQ_LEN), freqs).float()
    put(torch.cos(fr))
    put(torch.sin(fr))
    return np.concatenate(parts)


def hf_logits(hf, tokens) -> np.ndarray:
    with torch.no_grad():
        out = hf(torch.tensor([tokens], dtype=torch.long)).logits[0].float().numpy()
    return np.ascontiguousarray(out[list(KEEP)]).astype(np.float32)


def fixture_path(name: str) -> str:
    return os.path.join(HERE, "hf_full_" + name.replace(".", "_") + ".npz")


def make(name: str) -> None:  # build container only
    import tempfile
    REF = "/root/reference"
    sys.path.insert(0, os.path.join(REF, "tools"))
    exp = __import__("export" if CASES[name]["kind"] == "llama" else "export_qwen2")
    hf, tokens = build_hf(name)
    logits = hf_logits(hf, tokens)
    mine = export_image(hf, name)
    with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
        hf.save_pretrained(td)
        cwd = os.getcwd()
        os.chdir(td)  # load_hf_model reads ./config.json when present (tools/export.py:543-553)
        try:
            m = exp.load_hf_model(td)
        finally:
            os.chdir(cwd)
        p = os.path.join(td, "m.bin")
        exp.legacy_export(m, p)
        ref_bytes = np.fromfile(p, dtype=np.uint8)
    assert ref_bytes.size == mine.size and np.array_equal(ref_bytes, mine), \
        "export_image is not byte-identical to the reference exporter's file"
    sha = hashlib.sha256(ref_bytes.tobytes()).hexdigest()
    spec = spec_of(name)
    assert binfmt.image_nbytes(spec) == ref_bytes.size
    np.savez_compressed(fixture_path(name), tokens=np.asarray(tokens, np.int32), positions=np.asarray(KEEP, np.int32),
                        logits=logits, sha256=np.asarray(sha), image_bytes=np.asarray(ref_bytes.size, np.int64))
    print(f"wrote {fixture_path(name)}: image {ref_bytes.size} B sha256 {sha[:16]}..., logits {logits.shape}, "
          f"tokens {tokens}")


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES)):
        make(n)
