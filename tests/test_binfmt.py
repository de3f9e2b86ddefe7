"""Host logic: .bin layout table, header parsing, synthetic images. CPU only."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_MODELS, load_golden
from kuiperllama_amd import binfmt


@pytest.mark.parametrize("name", GOLDEN_MODELS)
def test_layout_matches_reference_exporter_bytes(name, oracle):
    """The layout table must account for every byte the REFERENCE exporter wrote."""
    spec, img, _, _ = load_golden(name)
    ents, total = binfmt.layout(spec)
    assert total == img.size
    # contiguous, ordered, no holes
    off = spec.header_bytes()
    for e in ents:
        assert e.offset == off
        off += e.nbytes
    assert off == total
    # header round trip
    s2 = binfmt.spec_from_image(img, family=spec.family, quant=spec.quant,
                                rope_mode=spec.rope_mode, rope_theta=spec.rope_theta,
                                rms_eps=spec.rms_eps)
    for f in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len",
              "shared_classifier"):
        assert getattr(s2, f) == getattr(spec, f)
    # the C oracle's independent offset walk agrees
    m = oracle.OracleModel.from_spec(img, spec)
    assert m.expected_bytes() == total


def test_freqs_block_matches_exporter():
    spec, img, _, _ = load_golden("ref_llama_gqa_tied")
    ents, _ = binfmt.layout(spec)
    e = {x.name: x for x in ents}
    cos = np.frombuffer(img.tobytes(), np.float32, count=spec.seq_len * spec.head_size // 2,
                        offset=e["freqs_cos"].offset)
    assert cos[0] == 1.0 and abs(cos[spec.head_size // 2] - np.cos(1.0)) < 1e-6


@pytest.mark.parametrize("preset,quant", [("stories15M", False)])
def test_synth_image_is_parseable(preset, quant, oracle):
    spec = binfmt.PRESETS[preset]
    img = binfmt.synth_image(spec, seed=7).numpy()
    assert img.size == binfmt.image_nbytes(spec) == 60816028
    m = oracle.OracleModel.from_spec(img, spec)
    lg = m.forward(1, 0)
    assert np.isfinite(lg).all() and lg.shape == (32000,)
    # determinism
    img2 = binfmt.synth_image(spec, seed=7).numpy()
    assert np.array_equal(img, img2)


def test_synth_int8_matches_oracle_quantizer(oracle):
    spec = binfmt.ModelSpec(128, 320, 2, 4, 4, 256, 32, False, binfmt.FAMILY_LLAMA, True, 64,
                            binfmt.ROPE_INTERLEAVED, 10000.0, 1e-5, "t")
    img = binfmt.synth_image(spec, seed=3)
    ents, total = binfmt.layout(spec)
    assert img.numel() == total
    w = torch.randn(64 * 33) * 0.02
    q, s = binfmt.quantize_q80_torch(w, 64)
    qo, so = oracle.quantize_q80(w.numpy(), 64)
    assert np.array_equal(q.numpy(), qo) and np.array_equal(s.numpy(), so)
    m = oracle.OracleModel.from_spec(img.numpy(), spec)
    assert np.isfinite(m.forward(3, 0)).all()


def test_rejects_broken_reference_combinations():
    bad = binfmt.ModelSpec(128, 320, 2, 4, 4, 256, 32, True, binfmt.FAMILY_LLAMA, True)
    with pytest.raises(ValueError):
        binfmt.layout(bad)  # int8 + tied classifier: llama3.cpp:259-262 is broken


def test_algorithmic_bytes_match_survey_table():
    # SURVEY.md §8(d) table
    s = binfmt.PRESETS["llama3.2-1b"]
    assert s.weight_elems() == 1_235_746_816
    assert abs(s.algorithmic_bytes_per_token(63.5) - 4.948e9) < 0.01e9
    s7 = binfmt.PRESETS["llama2-7b-int8"]
    assert s7.weight_elems() == 6_607_077_376
    assert abs(s7.algorithmic_bytes_per_token(63.5) - 7.088e9) < 0.01e9
    sq = binfmt.PRESETS["qwen2.5-0.5b"]
    assert sq.weight_elems() == 493_961_216
