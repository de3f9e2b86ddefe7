#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the REFERENCE's own Python
code (tools/model.py, tools/model_qwen2.py, tools/export.py, tools/export_qwen2.py) and, for
the rotate-half RoPE flavour, from HuggingFace's Llama/Qwen2 implementations exported through
the reference's ``load_hf_model``.

Runs ONLY in the build container (needs /root/reference).  Nothing in tests/ reads
/root/reference at run time: they read the .npz files this script wrote.

Each fixture = { image: the exact .bin bytes written by the reference exporter,
                 tokens: int32[T], logits: float32[T, V] (reference logits after feeding
                 tokens[0..t]), + the ModelSpec fields }.
"""
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "tools"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import export as ref_export  # noqa: E402  (reference exporter, llama)
import export_qwen2 as ref_export_qwen  # noqa: E402
import model as ref_model  # noqa: E402
import model_qwen2 as ref_model_qwen  # noqa: E402

from kuiperllama_amd import binfmt  # noqa: E402


def _logits_per_pos(fn, tokens):
    out = []
    with torch.no_grad():
        for t in range(len(tokens)):
            out.append(fn(torch.tensor([tokens[: t + 1]], dtype=torch.long)).reshape(-1).float().numpy())
    return np.stack(out).astype(np.float32)


def _save(name, image_path, tokens, logits, spec: binfmt.ModelSpec, extra=None):
    img = np.fromfile(image_path, dtype=np.uint8)
    d = dict(image=img, tokens=np.asarray(tokens, np.int32), logits=logits)
    for k, v in binfmt.spec_to_dict(spec).items():
        d["spec_" + k] = np.asarray(v)
    if extra:
        d.update(extra)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}: image {img.size} B, logits {logits.shape}")


def _randomise_norms(m, gen):
    for n, p in m.named_parameters():
        if n.endswith("norm.weight"):
            with torch.no_grad():
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen))


def make_ref_llama(name, *, dim, hidden, L, heads, kv_heads, vocab, seq_len, tied, seed, T=12):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    args = ref_model.ModelArgs(dim=dim, n_layers=L, n_heads=heads, n_kv_heads=kv_heads,
                               vocab_size=vocab, hidden_dim=hidden, max_seq_len=seq_len)
    m = ref_model.Transformer(args).eval()
    _randomise_norms(m, gen)
    if not tied:
        m.output.weight = torch.nn.Parameter(0.02 * torch.randn(vocab, dim, generator=gen))
        m.tok_embeddings.weight = torch.nn.Parameter(0.02 * torch.randn(vocab, dim, generator=gen))
    tokens = torch.randint(0, vocab, (T,), generator=gen).tolist()
    logits = _logits_per_pos(lambda tk: m(tk), tokens)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "m.bin")
        ref_export.legacy_export(m, p)
        spec = binfmt.ModelSpec(dim, hidden, L, heads, kv_heads, vocab, seq_len, tied,
                                binfmt.FAMILY_LLAMA, False, 64, binfmt.ROPE_INTERLEAVED, 10000.0,
                                1e-5, name)
        _save(name, p, tokens, logits, spec)
    return m, tokens


def make_ref_qwen_interleaved(name, *, dim, hidden, L, heads, kv_heads, vocab, seq_len, seed, T=12):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    args = ref_model_qwen.ModelArgs(dim=dim, n_layers=L, n_heads=heads, n_kv_heads=kv_heads,
                                    vocab_size=vocab, hidden_dim=hidden, max_seq_len=seq_len)
    m = ref_model_qwen.Transformer(args).eval()
    _randomise_norms(m, gen)
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n.endswith(".bias"):
                p_.copy_(0.02 * torch.randn(p_.shape, generator=gen))
    tokens = torch.randint(0, vocab, (T,), generator=gen).tolist()
    logits = _logits_per_pos(lambda tk: m(tk), tokens)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "m.bin")
        ref_export_qwen.legacy_export(m, p)
        spec = binfmt.ModelSpec(dim, hidden, L, heads, kv_heads, vocab, seq_len, True,
                                binfmt.FAMILY_QWEN2, False, 64, binfmt.ROPE_INTERLEAVED, 10000.0,
                                1e-5, name)
        _save(name, p, tokens, logits, spec)


def make_ref_int8(name, *, dim, hidden, L, heads, kv_heads, vocab, seq_len, seed, T=12):
    """int8: the reference has no CPU int8 backend; the golden logits come from the reference
    torch model run with DEQUANTISED weights (q * scale, tools/export.py:49-73)."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    args = ref_model.ModelArgs(dim=dim, n_layers=L, n_heads=heads, n_kv_heads=kv_heads,
                               vocab_size=vocab, hidden_dim=hidden, max_seq_len=seq_len)
    m = ref_model.Transformer(args).eval()
    _randomise_norms(m, gen)
    m.output.weight = torch.nn.Parameter(0.02 * torch.randn(vocab, dim, generator=gen))
    m.tok_embeddings.weight = torch.nn.Parameter(0.02 * torch.randn(vocab, dim, generator=gen))
    tokens = torch.randint(0, vocab, (T,), generator=gen).tolist()
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "m.bin")
        ref_export.legacy_export_quant(m, p)
        # dequantised twin for the golden logits
        with torch.no_grad():
            for n, w in m.named_parameters():
                if w.dim() == 2 and not n.startswith("tok_embeddings"):
                    q, s, _ = ref_export.quantize_q80(w, 64)
                    w.copy_((q.float() * s[:, None]).reshape(w.shape))
        logits = _logits_per_pos(lambda tk: m(tk), tokens)
        spec = binfmt.ModelSpec(dim, hidden, L, heads, kv_heads, vocab, seq_len, False,
                                binfmt.FAMILY_LLAMA, True, 64, binfmt.ROPE_INTERLEAVED, 10000.0,
                                1e-5, name)
        _save(name, p, tokens, logits, spec)


def make_hf(name, kind, *, dim, hidden, L, heads, kv_heads, vocab, seq_len, theta, eps, seed, T=12):
    """Rotate-half RoPE flavour (LLAMA3_SUPPORT / QWEN2_SUPPORT builds of the reference):
    HF random-init model -> reference load_hf_model -> reference legacy_export."""
    import transformers
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    common = dict(hidden_size=dim, intermediate_size=hidden, num_hidden_layers=L,
                  num_attention_heads=heads, num_key_value_heads=kv_heads, vocab_size=vocab,
                  max_position_embeddings=seq_len, rms_norm_eps=eps, rope_theta=theta,
                  tie_word_embeddings=True, initializer_range=0.02, attention_dropout=0.0)
    if kind == "llama":
        cfg = transformers.LlamaConfig(**common, rope_scaling=None, attention_bias=False,
                                       mlp_bias=False)
        hf = transformers.LlamaForCausalLM(cfg)
        exp = ref_export
        family = binfmt.FAMILY_LLAMA
    else:
        cfg = transformers.Qwen2Config(**common, use_sliding_window=False)
        hf = transformers.Qwen2ForCausalLM(cfg)
        exp = ref_export_qwen
        family = binfmt.FAMILY_QWEN2
    hf = hf.eval().float()
    with torch.no_grad():
        for n, p_ in hf.named_parameters():
            if n.endswith("norm.weight"):
                p_.copy_(1.0 + 0.1 * torch.randn(p_.shape, generator=gen))
            if n.endswith(".bias"):
                p_.copy_(0.02 * torch.randn(p_.shape, generator=gen))
    tokens = torch.randint(0, vocab, (T,), generator=gen).tolist()
    logits = _logits_per_pos(lambda tk: hf(tk).logits[:, -1, :], tokens)
    with tempfile.TemporaryDirectory() as td:
        hf.save_pretrained(td)
        cwd = os.getcwd()
        os.chdir(td)  # load_hf_model reads ./config.json when present (tools/export.py:543-553)
        try:
            m = exp.load_hf_model(td)
        finally:
            os.chdir(cwd)
        p = os.path.join(td, "m.bin")
        exp.legacy_export(m, p)
        spec = binfmt.ModelSpec(dim, hidden, L, heads, kv_heads, vocab, seq_len, True, family,
                                False, 64, binfmt.ROPE_HALF, theta, eps, name)
        _save(name, p, tokens, logits, spec)


def make_op_vectors():
    """Known-answer vectors held by the reference's own tests."""
    test_bin = np.fromfile(os.path.join(REF, "tmp", "test.bin"), dtype=np.uint8)
    np.savez_compressed(
        os.path.join(HERE, "ref_test_vectors.npz"),
        # test/test_op/test_cu_matmul.cpp:48-76
        matmul_x=np.array([1, 1, -1], np.float32),
        matmul_w=np.arange(1, 10, dtype=np.float32).reshape(3, 3),
        matmul_y=np.array([0, 3, 6], np.float32),
        # tmp/test.bin + test/test_op/test_load.cpp:11-108
        test_bin=test_bin,
        test_bin_header=np.array([16, 128, 256], np.int32),
        test_bin_matmul_idx=np.array([0, 1, 14, 15], np.int32),
        test_bin_matmul_out=np.array([8128, 24512, 237504, 253888], np.float32),
    )
    print("wrote ref_test_vectors.npz", test_bin.size)


if __name__ == "__main__":
    make_op_vectors()
    make_ref_llama("ref_llama_gqa_tied", dim=64, hidden=176, L=2, heads=4, kv_heads=2,
                   vocab=192, seq_len=32, tied=True, seed=11)
    make_ref_llama("ref_llama_mha_untied", dim=48, hidden=128, L=3, heads=6, kv_heads=6,
                   vocab=160, seq_len=24, tied=False, seed=12)
    make_ref_qwen_interleaved("ref_qwen_bias_interleaved", dim=64, hidden=160, L=2, heads=8,
                              kv_heads=2, vocab=200, seq_len=32, seed=13)
    make_ref_int8("ref_llama_int8_untied", dim=128, hidden=320, L=2, heads=4, kv_heads=4,
                  vocab=256, seq_len=32, seed=14)
    make_hf("hf_llama_half", "llama", dim=128, hidden=256, L=2, heads=4, kv_heads=2, vocab=224,
            seq_len=64, theta=500000.0, eps=1e-5, seed=15)
    make_hf("hf_qwen2_half", "qwen2", dim=128, hidden=192, L=2, heads=4, kv_heads=2, vocab=208,
            seq_len=64, theta=1000000.0, eps=1e-6, seed=16)
