"""The oracle (oracle/kuiper_oracle.c) against every golden vector the reference's own tests
hold for this path and against logits produced by the reference's Python
model/exporter (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, GOLDEN_MODELS, load_golden


def test_ref_matmul_known_answer(oracle):
    # test/test_op/test_cu_matmul.cpp:48-76: x=[1,1,-1], W=[[1..9]] -> [0,3,6]
    z = np.load(os.path.join(GOLDEN, "ref_test_vectors.npz"))
    y = oracle.matmul(z["matmul_x"], z["matmul_w"])
    assert np.array_equal(y, z["matmul_y"])
    y64 = oracle.matmul(z["matmul_x"], z["matmul_w"], acc=oracle.ACC_F64)
    assert np.array_equal(y64, z["matmul_y"])


def test_ref_test_bin_fixture(oracle):
    # tmp/test.bin; test/test_op/test_load.cpp:11-108
    z = np.load(os.path.join(GOLDEN, "ref_test_vectors.npz"))
    raw = z["test_bin"]
    hdr = raw[:28].view(np.int32)
    assert hdr[0] == 16 and hdr[1] == 128 and hdr[2] == 256
    w = raw[28:].view(np.float32)
    assert np.array_equal(w[:2048], np.arange(2048, dtype=np.float32))
    # CPU MatmulLayer 16x128 . ones (test_load.cpp:49-108)
    y = oracle.matmul(np.ones(128, np.float32), w[: 16 * 128].reshape(16, 128))
    assert np.array_equal(y[z["test_bin_matmul_idx"]], z["test_bin_matmul_out"])


def test_ref_embedding_arange(oracle):
    # test/test_op/test_cu_emb.cpp:6-89: table arange(4x512); token 1 -> 512+i, 2 -> 1024+i
    w = np.arange(4 * 512, dtype=np.float32).reshape(4, 512)
    out = oracle.embedding([1, 2], w)
    assert np.array_equal(out[0], 512 + np.arange(512, dtype=np.float32))
    assert np.array_equal(out[1], 1024 + np.arange(512, dtype=np.float32))
    with pytest.raises(IndexError):
        oracle.embedding([5], w)


def test_ref_add(oracle):
    # test/test_op/test_cu_add.cpp:7-75
    assert np.array_equal(oracle.add(np.full(4832, 2.0), np.full(4832, 3.0)), np.full(4832, 5.0, np.float32))
    np.testing.assert_allclose(oracle.add(np.full(62816, 2.1), np.full(62816, 3.3)), 5.4, rtol=1e-6)


def test_rmsnorm_swiglu_softmax_vs_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in (32, 480, 72480):  # sizes of test/test_op/test_cu_rmsnorm.cpp
        x = rng.random(n, dtype=np.float32)
        w = rng.random(n, dtype=np.float32)
        ref = w.astype(np.float64) * (x / np.sqrt(np.mean(x.astype(np.float64) ** 2) + 1e-5))
        np.testing.assert_allclose(oracle.rmsnorm(x, w, 1e-5), ref, rtol=0, atol=1e-5)
    a = rng.standard_normal(4832).astype(np.float32)
    b = rng.standard_normal(4832).astype(np.float32)
    ref = a.astype(np.float64) / (1 + np.exp(-a.astype(np.float64))) * b
    np.testing.assert_allclose(oracle.swiglu(a, b), ref, atol=1e-5)
    s = oracle.softmax(a)
    e = np.exp(a.astype(np.float64) - a.max())
    np.testing.assert_allclose(s, e / e.sum(), atol=1e-7)
    assert oracle.argmax(np.array([1, 7, 7, 3], np.float32)) == 1  # first maximum


def test_rope_modes_vs_formula(oracle):
    hs, dim, kv = 8, 32, 16
    rng = np.random.default_rng(1)
    q = rng.standard_normal(dim).astype(np.float32)
    k = rng.standard_normal(kv).astype(np.float32)
    for theta in (10000.0, 500000.0):
        s, c = oracle.sincos_cache(hs, 16, theta)
        d = np.arange(hs, dtype=np.float32)
        freq = (1.0 / np.power(np.float32(theta), d / np.float32(hs))).astype(np.float32)
        np.testing.assert_allclose(s[5], np.sin(np.float32(5) * freq), atol=1e-6)
        pos = 5
        # interleaved (cpu/rope_kernel.cpp:98-121)
        qi, ki = oracle.rope(q, k, pos, s, c, hs, oracle.ROPE_INTERLEAVED)
        for vec, out, n in ((q, qi, dim), (k, ki, kv)):
            for i in range(0, n, 2):
                fc, fs = c[pos, i % hs], s[pos, i % hs]
                np.testing.assert_allclose(out[i], vec[i] * fc - vec[i + 1] * fs, atol=1e-6)
                np.testing.assert_allclose(out[i + 1], vec[i] * fs + vec[i + 1] * fc, atol=1e-6)
        # half (cpu/rope_kernel.cpp:18-42)
        qh, kh = oracle.rope(q, k, pos, s, c, hs, oracle.ROPE_HALF)
        for vec, out, n in ((q, qh, dim), (k, kh, kv)):
            for h0 in range(0, n, hs):
                for j in range(hs // 2):
                    fc, fs = c[pos, 2 * j], s[pos, 2 * j]
                    a, b = vec[h0 + j], vec[h0 + j + hs // 2]
                    np.testing.assert_allclose(out[h0 + j], a * fc - b * fs, atol=1e-6)
                    np.testing.assert_allclose(out[h0 + j + hs // 2], a * fs + b * fc, atol=1e-6)


def test_quantizer_matches_export_semantics(oracle):
    # tools/export.py:49-73: scale = max|w|/127, q = round(w/scale), dequant error <= scale/2
    rng = np.random.default_rng(2)
    w = (0.02 * rng.standard_normal(64 * 50)).astype(np.float32)
    q, s = oracle.quantize_q80(w, 64)
    assert q.min() >= -127 and q.max() <= 127
    wg = w.reshape(-1, 64)
    np.testing.assert_allclose(s, np.abs(wg).max(1) / np.float32(127.0), rtol=1e-7)
    deq = q.reshape(-1, 64).astype(np.float32) * s[:, None]
    assert np.all(np.abs(deq - wg) <= s[:, None] * 0.5 + 1e-9)


@pytest.mark.parametrize("name", GOLDEN_MODELS)
def test_model_logits_vs_reference_python(oracle, name):
    """Whole-model pin: logits after each of 12 tokens vs the reference's Python model run on
    the .bin written by the reference's exporter.  fp32 tolerance 2e-6 abs (logits ~ O(1))."""
    spec, img, toks, ref = load_golden(name)
    m = oracle.OracleModel.from_spec(img, spec)
    assert m.expected_bytes() == img.size
    for acc in (oracle.ACC_F32, oracle.ACC_F64):
        m2 = oracle.OracleModel.from_spec(img, spec)
        for t, tok in enumerate(toks):
            lg = m2.forward(int(tok), t, acc)
            np.testing.assert_allclose(lg, ref[t], rtol=0, atol=2e-6)
            assert int(np.argmax(lg)) == int(np.argmax(ref[t]))


def test_generate_loop_semantics(oracle):
    """demo/main.cpp:5-47: prompt tokens are forced, then greedy; words has total_steps entries."""
    spec, img, toks, ref = load_golden("ref_llama_gqa_tied")
    m = oracle.OracleModel.from_spec(img, spec)
    prompt = [int(t) for t in toks[:3]]
    words = m.generate(prompt, 10)
    assert len(words) == 10
    assert words[:2] == prompt[1:3]           # forced while pos < prompt_len-1
    # step 2 is the first sampled one: argmax of the reference logits after 3 prompt tokens
    assert words[2] == int(np.argmax(ref[2]))
    # replay by hand
    m2 = oracle.OracleModel.from_spec(img, spec)
    seq = list(prompt)
    for pos in range(10):
        lg = m2.forward(seq[pos], pos)
        if pos + 1 >= len(seq):
            seq.append(int(np.argmax(lg)))
    assert seq[1:11] == words


def test_oracle_generate_stop_semantics(oracle):
    """demo/main.cpp:30-32: break on a sampled stop token before it is appended."""
    spec, img, toks, _ = load_golden("ref_llama_gqa_tied")
    om = oracle.OracleModel.from_spec(img, spec)
    prompt = [int(t) for t in toks[:3]]
    full = om.generate(prompt, 24)
    assert full[:2] == prompt[1:]
    k = 9
    got = om.generate(prompt, 24, stop=[full[k]])
    first = next(i for i in range(2, 24) if full[i] == full[k])
    assert got == full[:first]
    assert om.generate(prompt, 24, stop=[prompt[1]])[:2] == prompt[1:]  # prompt ids never stop it
